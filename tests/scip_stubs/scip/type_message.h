/* test-only stand-in (tests/scip_stubs/README.md) */
#ifndef HIPSDP_TEST_STUB_TYPE_MESSAGE_H
#define HIPSDP_TEST_STUB_TYPE_MESSAGE_H
typedef struct SCIP_Messagehdlr SCIP_MESSAGEHDLR;
#endif
