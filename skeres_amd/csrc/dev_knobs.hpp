// The developer variables of libskeres_amd — one struct, one reader (dev_knobs(), capi.hip).
#pragma once

namespace sk {

// The developer variables of the library: ALL of them, read ONCE, here (dev_knobs(), capi.hip).  None is needed in production
// and none changes a result beyond rounding; what a caller may legitimately choose goes through sk_options_set_* (the block
// envelope, the dissection, the border, the SYRK grouping and look-ahead, resident kernels on or off).  Round 3 had 44 such
// variables scattered over the sources; the tuning sweeps they served are recorded in DESIGN.md and profiles/, their
// settled values are constants now.
struct DevKnobs {
  int chain_server = 1;            // SK_CHOL_CHAIN_SERVER=0: no resident panel chain in this process (the same plan, launch by launch) — counter-collection runs
  int bs_resident = 1;             // SK_BS_RESIDENT=0: the back-substitution as one launch per block step (the bitwise comparison of the two forms)
  const char* chain_stamps = nullptr;  // SK_CHAIN_STAMPS=<file>: device time stamps of the resident chain (tools/chain_timeline.py)
  const char* bs_stamps = nullptr;     // SK_BS_STAMPS=<file>: ... of the resident back-substitution
  bool debug_queues = false, debug_envelope = false, debug_segments = false, debug_chain_abort = false, debug_setup = false;  // SK_DEBUG=queues,envelope,segments,chain_abort,setup: lines on stderr
  int queue_shift = 0;             // SK_QUEUE_SHIFT=<n>: creates n hardware queues first, as another library in the process would (queue-placement tests)
  int chain_queues = -1;           // SK_CHAIN_QUEUES=<k>: fixes the (bulk, panel, server) queue combination instead of the trial run
  int pair_max_trailing = 0;       // SK_CHAIN_PAIR_MAX_TRAILING=<rows>: the resident-pairs plan of round 3 (off by default; tests/pair_plan_worker.py)
  int dissect_at = -1;             // SK_DISSECT_AT=<cameras>: head size of a forced two-way dissection (0: none) — cut sweeps
  int chain_early_server = 1;  // SK_CHAIN_EARLY_SERVER=0: the potrf server's launch behind an event of the caller's stream (until round 5) instead of waiting in the kernel for chain_start_kernel
  int bs_spread = 1;  // SK_BS_SPREAD=0: the resident back-substitution's owners on consecutive workgroups (eight XCDs in turn) and every hop through memory, as until round 5
  int bs_pair = 1;  // SK_BS_PAIR=0: the two leaf fronts' back-substitutions as two launches on two streams (until round 5) instead of one launch
  int bulk_reserve = -1, bulk_reserve_early = -1;  // SK_BULK_RESERVE=<a>[,<b>]: CUs per XCD kept free of the SYRKs on the bulk / early-bulk streams (sweeps; default 4, 2)
  int chain_xcd_local = 0;         // SK_CHAIN_XCD_LOCAL=1: the resident chain's critical hand-overs through the potrf server's XCD's L2 instead of device-scope counters (round 5: pays on one front, not on two in lock-step)
  int schedule_plain = 0;          // SK_SCHEDULE_PLAIN=1: the envelope zeroed in line and the pair kernels in plain block order (the schedule-independence test)
};
const DevKnobs& dev_knobs();

}  // namespace sk
