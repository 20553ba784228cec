// comm_schedule.hpp -- the stream / event schedule of the PIPELINED table reduce (kpal_comm_reduce_table_async, kpal_multi.hip),
// written against an abstract runtime so that the same code that drives HIP streams and RCCL in the library can be driven by
// a fake runtime on the CPU (tests/native/comm_schedule_check.cpp: two ranks as threads, streams as worker threads, events as
// flags, the reduce as a rendezvous that adds the ranks' buffers) under ThreadSanitizer -- no multi-GPU box is needed to find a
// missing wait in the side-buffer alternation.
//
// The schedule.  Step i's table is copied to side buffer t = i mod 2 on the MAIN stream and reduced (+ balanced on the root)
// THERE on the COMMUNICATOR's stream, while the main stream goes on with the count of step i + 1 into the table:
//     main:  [count i] copy table -> side[t]  record(copied)          [count i + 1] ...
//     comm:                                   wait(copied)  reduce(side[t])  balance(side[t])  record(side_free[t])
// side[t]'s previous content (the merged table of step i - 2) may be overwritten once ITS reduce + balance are done: the main
// stream waits for side_free[t] before the copy.  A side buffer that must grow is freed only after the host has waited for that
// event.  `merged` names the buffer the merged table of the last issued step lies in (valid once the communicator's stream has
// caught up; kpal_sync waits for both streams).
//
// Runtime R provides (all return 0 or an error code that ends the schedule):
//   size_t side_capacity(int t);  int side_grow(int t, size_t bytes);          // (grow: free + allocate; content lost)
//   int host_wait_side_free(int t);  int main_wait_side_free(int t);
//   int main_copy_table_to_side(int t, size_t bytes);  int main_record_copied();  int comm_wait_copied();
//   int comm_reduce_side(int t, int root);  int comm_balance_side(int t);  int comm_record_side_free(int t);
//   void *side_ptr(int t);
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace kpal {

struct CommPipeState {
    int side_turn = 0;
    bool side_used[2] = {false, false};
    void *merged = nullptr;
    uint64_t merged_bins = 0, merged_first = 0;
};

template <class R>
int comm_reduce_async_schedule(R &rt, CommPipeState &st, uint64_t bins, int rank, int root, bool balance)
{
    st.side_turn ^= 1;
    const int t = st.side_turn;
    const size_t bytes = (size_t)bins * sizeof(int64_t);
    if (rt.side_capacity(t) < bytes) {
        if (st.side_used[t]) {   // (re-allocation: its last reader -- on the communicator's stream -- must be done)
            if (int rc = rt.host_wait_side_free(t)) return rc;
            st.side_used[t] = false;
            if (st.merged == rt.side_ptr(t)) st.merged = nullptr, st.merged_bins = 0;
        }
        if (int rc = rt.side_grow(t, bytes)) return rc;
    }
    // the buffer's previous content (the merged table of two steps ago) may go once its reduce + balance are done
    if (st.side_used[t])
        if (int rc = rt.main_wait_side_free(t)) return rc;
    if (int rc = rt.main_copy_table_to_side(t, bytes)) return rc;
    if (int rc = rt.main_record_copied()) return rc;
    if (int rc = rt.comm_wait_copied()) return rc;
    if (int rc = rt.comm_reduce_side(t, root)) return rc;
    if (balance && rank == root)
        if (int rc = rt.comm_balance_side(t)) return rc;
    if (int rc = rt.comm_record_side_free(t)) return rc;
    st.side_used[t] = true;
    st.merged = rt.side_ptr(t);
    st.merged_bins = bins;
    st.merged_first = 0;
    return 0;
}

}  // namespace kpal
