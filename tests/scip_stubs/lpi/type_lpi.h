/* test-only stand-in (tests/scip_stubs/README.md) */
#ifndef HIPSDP_TEST_STUB_TYPE_LPI_H
#define HIPSDP_TEST_STUB_TYPE_LPI_H
#endif
