/* test-only stand-in (tests/scip_stubs/README.md) */
#ifndef HIPSDP_TEST_STUB_PUB_MESSAGE_H
#define HIPSDP_TEST_STUB_PUB_MESSAGE_H
#include "scip/type_message.h"
void SCIPmessagePrintError(const char* formatstr, ...);
void SCIPmessagePrintInfo(SCIP_MESSAGEHDLR* messagehdlr, const char* formatstr, ...);
#define SCIPerrorMessage SCIPmessagePrintError
#endif
