/* hipsdp_scip_compat.h - the handful of SCIP / SCIP-SDP types that the solver-interface boundary mentions, for building
 * sdpisolver_hip.c and lapack_interface_hip.c WITHOUT a SCIP installation (tests, bench, this repository's CI).
 *
 * When the backend is built inside a SCIP-SDP tree (-DHIPSDP_WITH_SCIP) none of this is used: the real headers
 * scip/def.h, blockmemshell/memory.h, scip/type_retcode.h, scip/type_message.h, sdpi/type_sdpi.h, sdpi/sdpiclock.h are
 * included instead (reference: src/sdpi/sdpisolver.h:57-62).  Names and numeric values below follow those headers so that
 * the same object code semantics hold in both builds; nothing here is an implementation copied from SCIP.
 */
#ifndef HIPSDP_SCIP_COMPAT_H
#define HIPSDP_SCIP_COMPAT_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef double       SCIP_Real;
typedef unsigned int SCIP_Bool;
#ifndef TRUE
#define TRUE  1
#define FALSE 0
#endif
#define SCIP_EXPORT __attribute__((visibility("default")))
#define REALABS(x) ((x) < 0 ? -(x) : (x))

/* return codes: the values of SCIP's type_retcode.h (only the ones the SDPI layer distinguishes) */
enum SCIP_Retcode
{
   SCIP_OKAY              =  +1,
   SCIP_ERROR             =   0,
   SCIP_NOMEMORY          =  -1,
   SCIP_READERROR         =  -2,
   SCIP_WRITEERROR        =  -3,
   SCIP_NOFILE            =  -4,
   SCIP_FILECREATEERROR   =  -5,
   SCIP_LPERROR           =  -6,
   SCIP_NOPROBLEM         =  -7,
   SCIP_INVALIDCALL       =  -8,
   SCIP_INVALIDDATA       =  -9,
   SCIP_INVALIDRESULT     = -10,
   SCIP_PLUGINNOTFOUND    = -11,
   SCIP_PARAMETERUNKNOWN  = -12,
   SCIP_PARAMETERWRONGTYPE = -13,
   SCIP_PARAMETERWRONGVAL = -14,
   SCIP_NOTIMPLEMENTED    = -18
};
typedef enum SCIP_Retcode SCIP_RETCODE;

/* opaque in the standalone build: allocation goes through hipsdp_compat_malloc/free, which count live bytes so that the
 * tests can repeat the reference's leak check (unittests/src/checksdpi.c:117) */
typedef struct BMS_BlkMem BMS_BLKMEM;
typedef struct BMS_BufMem BMS_BUFMEM;
typedef struct SCIP_Messagehdlr SCIP_MESSAGEHDLR;

void*     hipsdp_compat_malloc(size_t bytes);
void*     hipsdp_compat_realloc(void* p, size_t oldbytes, size_t newbytes);
void      hipsdp_compat_free(void* p, size_t bytes);
long long hipsdp_compat_mem_used(void);
long long hipsdp_compat_shortcut_checks(void);      /* checks done under HIPSDP_VERIFY_SHORTCUT=1 (sdpisolver_hip.c) */

/* parameter ids and settings: values of src/sdpi/type_sdpi.h:47-79 */
enum SCIP_SDPParam
{
   SCIP_SDPPAR_EPSILON          = 0,
   SCIP_SDPPAR_GAPTOL           = 1,
   SCIP_SDPPAR_FEASTOL          = 2,
   SCIP_SDPPAR_SDPSOLVERFEASTOL = 3,
   SCIP_SDPPAR_OBJLIMIT         = 4,
   SCIP_SDPPAR_SDPINFO          = 5,
   SCIP_SDPPAR_SLATERCHECK      = 6,
   SCIP_SDPPAR_PENALTYPARAM     = 7,
   SCIP_SDPPAR_MAXPENALTYPARAM  = 8,
   SCIP_SDPPAR_NPENALTYINCR     = 9,
   SCIP_SDPPAR_LAMBDASTAR       = 10,
   SCIP_SDPPAR_NTHREADS         = 11,
   SCIP_SDPPAR_WARMSTARTPOGAP   = 12,
   SCIP_SDPPAR_PENINFEASADJUST  = 13,
   SCIP_SDPPAR_USEPRESOLVING    = 14,
   SCIP_SDPPAR_USESCALING       = 15,
   SCIP_SDPPAR_SCALEOBJ         = 16
};
typedef enum SCIP_SDPParam SCIP_SDPPARAM;

enum SCIP_SDPSolverSetting
{
   SCIP_SDPSOLVERSETTING_UNSOLVED = -1,
   SCIP_SDPSOLVERSETTING_PENALTY  = 0,
   SCIP_SDPSOLVERSETTING_FAST     = 1,
   SCIP_SDPSOLVERSETTING_MEDIUM   = 2,
   SCIP_SDPSOLVERSETTING_STABLE   = 3
};
typedef enum SCIP_SDPSolverSetting SCIP_SDPSOLVERSETTING;

/* clock handed in by the caller (src/sdpi/sdpiclock.h:50-78); standalone implementation in compat/sdpiclock_compat.c */
enum SDPI_ClockType { SDPI_CLOCKTYPE_CPU = 1, SDPI_CLOCKTYPE_WALL = 2 };
typedef enum SDPI_ClockType SDPI_CLOCKTYPE;
typedef struct SDPI_Clock SDPI_CLOCK;

SCIP_RETCODE SDPIclockCreate(SDPI_CLOCK** clck);
void         SDPIclockFree(SDPI_CLOCK** clck);
void         SDPIclockSetType(SDPI_CLOCK* clck, SDPI_CLOCKTYPE clocktype);
void         SDPIclockStart(SDPI_CLOCK* clck);
void         SDPIclockStop(SDPI_CLOCK* clck);
SCIP_Real    SDPIclockGetTime(SDPI_CLOCK* clck);

#ifdef __cplusplus
}
#endif

#endif
