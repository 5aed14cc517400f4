/* hs_util.cpp - error bookkeeping shared by the device engine */
#include "hs_common.h"
#include <stdio.h>
#include <string.h>

static thread_local char hs_errbuf[512] = "";

void hs_record_hip_error(hipError_t e, const char* what, const char* file, int line)
{
   snprintf(hs_errbuf, sizeof(hs_errbuf), "%s:%d: %s -> %s", file, line, what, hipGetErrorString(e));
}

const char* hs_last_error(void)
{
   return hs_errbuf;
}
