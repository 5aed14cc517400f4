/* solve1_c16.hip - the one-launch node solve (csrc/solve1_body.h), instance for problems whose blocks all have at most 16 rows, m <= 64 */
#define S1_NCLS 16
#define S1_MBIG 0
#define S1_KERNEL k_solve1_c16
#define S1_LAUNCH hs_solve1_launch_c16
#include "solve1_body.h"
