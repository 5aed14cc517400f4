/* solve1_c10.hip - the one-launch node solve (csrc/solve1_body.h), instance for problems whose blocks all have at most 10 rows, m <= 64 */
/* 256 threads (one wavefront per SIMD, 512 registers each): the register forms of the 10-row blocks keep 55 doubles of a matrix per
 * lane - with 512 threads (256 registers) this instance spills 347 vector registers into scratch memory, with 256 none */
#define S1_NT 256
#define S1_NW 4
#define S1_NCLS 10
#define S1_MBIG 0
#define S1_KERNEL k_solve1_c10
#define S1_LAUNCH hs_solve1_launch_c10
#include "solve1_body.h"
