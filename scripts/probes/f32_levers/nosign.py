# the forward-side twin: no sign-bit packing in the training forward's epilogue (timing only)
SUBS = [("""                    if (MODE == kTrainFwd)
                        sign_bits[o >> 1] = __builtin_amdgcn_alignbit(sign_bits[o >> 1],
                                                                      __builtin_bit_cast(unsigned, 0.0f - t), 31);""", "")]
