# lever (b), upper bound: backward data WITHOUT applying the ReLU masks at all (wrong results,
# timing only): what a free mask application would be worth
SUBS = [("""                    y[p] = __builtin_bit_cast(float, __builtin_bit_cast(int, a) & keep);   // element reads lane 0)""",
         """                    y[p] = a; (void)keep;""")]
