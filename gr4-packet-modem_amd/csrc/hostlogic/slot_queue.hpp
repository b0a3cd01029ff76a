// hostlogic/slot_queue.hpp -- the hand-over of batch slots between the stage threads of the receivers
// (csrc/packet_receiver.hip, csrc/multichannel_receiver.hip).  No HIP, no allocation: this header also compiles with
// plain g++ into the sanitizer targets of tests/hostlogic/ (ASan / UBSan / TSan; `make -C tests/hostlogic SAN=thread`).
//
// A receiver owns a fixed set of slots (one batch each).  Slot indices travel through the stages in submission order:
//   caller --submit--> q[0] -> stage thread 0 -> q[1] -> stage thread 1 -> ... -> done --collect--> caller
// push() never allocates and never throws (a std::deque did both: an exception between two stages would have left
// the batch in no queue, and collect() waiting for ever).
#pragma once
#include <condition_variable>
#include <cstddef>
#include <mutex>

namespace gr4pm {
namespace hostlogic {

template <int CAP>
class SlotQueue {
    int ring_[CAP];
    int head_ = 0, count_ = 0;
    bool quit_ = false;
    mutable std::mutex m_;
    std::condition_variable cv_;

public:
    static constexpr int kQuit = -1;
    // false (and nothing queued) when the ring is full: more slots in flight than the receiver owns -- a logic error of
    // the caller, reported instead of overwriting an entry
    bool push(int v) noexcept
    {
        {
            std::lock_guard<std::mutex> l(m_);
            if (count_ == CAP) return false;
            ring_[(head_ + count_) % CAP] = v;
            ++count_;
        }
        cv_.notify_all();
        return true;
    }
    // blocks; kQuit after stop() once the ring has drained (queued slots are still delivered), or when a kQuit
    // sentinel was pushed
    int pop() noexcept
    {
        std::unique_lock<std::mutex> l(m_);
        cv_.wait(l, [&] { return quit_ || count_ > 0; });
        if (count_ == 0) return kQuit;
        const int v = ring_[head_];
        head_ = (head_ + 1) % CAP;
        --count_;
        return v;
    }
    void stop() noexcept
    {
        {
            std::lock_guard<std::mutex> l(m_);
            quit_ = true;
        }
        cv_.notify_all();
    }
    size_t size() const noexcept
    {
        std::lock_guard<std::mutex> l(m_);
        return static_cast<size_t>(count_);
    }
};

// Body of one stage thread: pops slot indices from `from` until kQuit, runs body(slot) under a catch-all (an exception
// becomes on_fail(slot, what) -- the batch is marked failed and still travels on, so collect() gets it) and hands the
// slot to `to`.  When the input ends, the end travels on too (`to->push(kQuit)`), so one stop() at the head of a chain
// winds the whole chain down in order.
template <class Q, class Body, class Fail>
void run_stage(Q& from, Q* to, bool forward_quit, Body&& body, Fail&& on_fail) noexcept
{
    for (;;) {
        const int slot = from.pop();
        if (slot < 0) break;
        try {
            body(slot);
        } catch (...) {
            on_fail(slot);
        }
        if (to) (void)to->push(slot);
    }
    if (to && forward_quit) (void)to->push(Q::kQuit);
}

} // namespace hostlogic
} // namespace gr4pm
