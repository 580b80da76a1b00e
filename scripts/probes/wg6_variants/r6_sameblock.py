# timing-only knock-out (WRONG results): every LDS-DMA stage of wgrad_bf16x6.hip requests the segment's FIRST block
# (the requests, their LDS writes and waits stay; the data comes out of L2, not HBM)
SUBS = {"wgrad_bf16x6.hip": [("""        const int64_t blk = seg.blk_begin + (st >> 1);
        const int half = (int)(st & 1) * 256;""", """        const int64_t blk = seg.blk_begin + ((st >> 1) & 3);
        const int half = (int)(st & 1) * 256;""")]}
