// decompress_rows_m1.hip -- the fp32-arithmetic (simulated path) instantiations of the row decompressor (decompress_rows_impl.h)
#define DEC_PART 1
#define DEC_ENTRY gear_decompress_rows_m1
#include "decompress_rows_impl.h"
