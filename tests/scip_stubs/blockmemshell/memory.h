/* test-only stand-in (tests/scip_stubs/README.md) */
#ifndef HIPSDP_TEST_STUB_MEMORY_H
#define HIPSDP_TEST_STUB_MEMORY_H
#include <stddef.h>
typedef struct BMS_BlkMem BMS_BLKMEM;
typedef struct BMS_BufMem BMS_BUFMEM;
void* BMSallocBlockMemory_call(BMS_BLKMEM* blkmem, size_t size);
void  BMSfreeBlockMemory_call(BMS_BLKMEM* blkmem, void** ptr, size_t size);
#define BMSallocBlockMemory(mem, ptr)                 (*(void**) (ptr) = BMSallocBlockMemory_call((mem), sizeof(**(ptr))))
#define BMSallocBlockMemoryArray(mem, ptr, num)       (*(void**) (ptr) = BMSallocBlockMemory_call((mem), sizeof(**(ptr)) * (size_t) (num)))
#define BMSfreeBlockMemory(mem, ptr)                  BMSfreeBlockMemory_call((mem), (void**) (ptr), sizeof(**(ptr)))
#define BMSfreeBlockMemoryArrayNull(mem, ptr, num)    do { if ( *(ptr) != NULL ) BMSfreeBlockMemory_call((mem), (void**) (ptr), sizeof(**(ptr)) * (size_t) (num)); } while (0)
#endif
