# every weight request issued twice (second copy into a scratch register that is kept alive): what
# would a 16-wave organisation with two waves per output tile cost on the L1 / L2 path?
SUBS = [("""        for (int part = 0; part < 2; ++part)
            dst[t][part] = __builtin_bit_cast(bf16x8, base[part * 64 + w.lane]);
    }
}""", """        for (int part = 0; part < 2; ++part) {
            dst[t][part] = __builtin_bit_cast(bf16x8, base[part * 64 + w.lane]);
            f32x4 dup = __builtin_nontemporal_load(&base[part * 64 + w.lane]);
            asm volatile("" ::"v"(dup));
        }
    }
}""")]
