/* Entry points that exist only in the TEST build of the library (gr4-packet-modem_amd/libgr4pm_hip_test.so: the
 * same objects with csrc/common.hip compiled under -DGR4PM_TEST_ALLOC_HOOK; `make -C gr4-packet-modem_amd/csrc
 * test_lib`, built by __graft_entry__.build()).  The shipped libgr4pm_hip.so has neither these symbols nor an
 * operator new of its own. */
#pragma once
#ifdef __cplusplus
extern "C" {
#endif
/* Test-only fault injection (tests/test_abi_and_host.py, tests/test_gpu_parity.py): the library's own operator new
 * lets `after` allocations pass, fails the next `count` (std::bad_alloc) and disarms itself; after < 0 disarms.  Only
 * allocations made by this library's code are affected (the operator has hidden visibility).  What is being tested is
 * the first convention above: the failure must come back as GR4PM_ERR_NOMEM from the entry point -- or as the failed
 * batch's status when it strikes in one of a receiver's stage threads -- never as an exception or a std::terminate. */
void gr4pm_test_fail_allocations(long after, long count);
unsigned long long gr4pm_test_allocation_count(void);
#ifdef __cplusplus
}
#endif
