SUBS = [("""                if (c0 + G < g_trig) features16<true>(enc, c0 + G, w.h, p0, p1, p2, f);
                else features16<false>(enc, c0 + G, w.h, p0, p1, p2, f);""",
         """#pragma unroll
                for (int j = 0; j < 8; ++j) f[j] = p0 + (float)(c0 + G + j);""")]
