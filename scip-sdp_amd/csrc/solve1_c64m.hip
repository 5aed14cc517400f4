/* solve1_c64m.hip - the one-launch node solve (csrc/solve1_body.h), instance for 64 < m <= 128 */
#define S1_NCLS 64
#define S1_MBIG 1
#define S1_KERNEL k_solve1_c64m
#define S1_LAUNCH hs_solve1_launch_c64m
#include "solve1_body.h"
