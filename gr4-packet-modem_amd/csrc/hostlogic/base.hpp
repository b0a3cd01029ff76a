// hostlogic/base.hpp -- what the HIP-free host state machines of the library share.
//
// Everything under csrc/hostlogic/ is the PRODUCT's own host logic (the tag / message state machines the reference
// blocks run per processBulk() call, replayed over a whole call here; the slot rings of the receivers), written so that
// it compiles with plain g++ and no HIP header: the .hip files include these headers and launch kernels from what they
// return, and tests/hostlogic/ builds exactly the same code with -fsanitize=address / undefined / thread and checks
// it against the CPU oracle (`make -C tests/hostlogic SAN=...`, run by tests/test_hostlogic_sanitizers.py).
#pragma once
#include <cstddef>
#include <cstdint>

#include "../../../include/gr4pm_hip.h"

namespace gr4pm {

void set_error(const char* fmt, ...); // csrc/common.hip (the library), tests/hostlogic/hostlogic_san.cpp (the sanitizer build)

namespace hostlogic {

// `len` items from in[src] to out[dst]: what the state machines that only move items (SyncwordDetectionFilter,
// PayloadMetadataInsert, SyncwordRemove, HeaderPayloadSplit) produce instead of moving anything themselves
struct CopySpan {
    unsigned long long src, dst, len;
};

} // namespace hostlogic
} // namespace gr4pm
