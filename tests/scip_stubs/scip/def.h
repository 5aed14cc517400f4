/* test-only stand-in (tests/scip_stubs/README.md) */
#ifndef HIPSDP_TEST_STUB_SCIP_DEF_H
#define HIPSDP_TEST_STUB_SCIP_DEF_H
#include <stdio.h>
#include <limits.h>
#include <assert.h>
#define SCIP_EXPORT
#define SCIP_Bool unsigned int
#ifndef TRUE
#define TRUE 1
#define FALSE 0
#endif
#define SCIP_Real double
#define SCIP_Longint long long
#define REALABS(x) (fabs(x))
#define MAX(x, y) ((x) >= (y) ? (x) : (y))
#define MIN(x, y) ((x) <= (y) ? (x) : (y))
#define SCIP_CALL(x) do { SCIP_RETCODE rc_; if ( (rc_ = (x)) != SCIP_OKAY ) return rc_; } while (0)
#define SCIPABORT() assert(0)
#endif
