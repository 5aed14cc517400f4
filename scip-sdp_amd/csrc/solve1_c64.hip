/* solve1_c64.hip - the one-launch node solve (csrc/solve1_body.h), instance for any block sizes, m <= 64 */
#define S1_NCLS 64
#define S1_MBIG 0
#define S1_KERNEL k_solve1_c64
#define S1_LAUNCH hs_solve1_launch_c64
#include "solve1_body.h"
