// tools/archive/zf_persist_kernel.h - ARCHIVED EXPERIMENT (round 4), not part of the library.
//
// Several full-chain passes of the separable solver in ONE launch, with an in-kernel decide and a descriptor barrier.
// Built, proven bit-identical to one launch per pass, measured at every size - and slower or equal everywhere
// (profiles/r04_persist_vs_per_pass_by_size.jsonl, profiles/r04_persist_phases_*.txt; HISTORY.md, round 4).  Round 5
// removed it from the product (kernel, translation unit, ZF_PERSIST* switches, its host code in zf_solver.hip and
// tests/test_gpu_persist.py).  The text below is the kernel as it last shipped; it compiles against
// zfista_amd/csrc/zf_kernels_step.h of commit dc387b1 (`git show dc387b1:zfista_amd/csrc/zf_solver.hip` has the host
// side: zf_persist_run, zf_launch_persist; `git show dc387b1:tests/test_gpu_persist.py` its tests).
#pragma once
// ---------------------------------------------------------------------------
// Several passes in ONE launch (grids whose workgroups are all resident at once: n up to ~2.5e7).
// A per-pass launch costs what lies between two dependent kernels of a stream (8 - 10 us) and the ramp of its one
// round of workgroups; at n = 1e7 that is a quarter of a 0.15 ms pass, at n <= 1e6 most of it.  This kernel runs up
// to `npass` FULL-CHAIN passes back to back: every workgroup keeps its tiles, the last arriver of a pass decides it as
// zf_pass_tail always does, and instead of ending the launch it publishes the new control block; the others wait for
// its sequence number and go on.  It leaves - before touching anything - as soon as the next pass is not a full chain
// (a chain broke, the tail before max_iter, a final status): the per-pass kernels the host enqueues behind it take over,
// so it needs no other body than the hot one (all bodies in one kernel: 274+ VGPRs, one wave per SIMD).
//
// Memory model.  The L2s of the 8 XCDs are not coherent with each other and a launch is no longer a boundary, so
// inside this kernel the control block is only ever read and written PAST the caches: every workgroup fetches its own
// copy (52 words, one agent-scope load per lane) at the start of a pass; the deciding wave runs zf_decide_pass on its
// copy in LDS and publishes it word by word, the word that carries pass_seq last, behind a counted wait.  The
// iterates a workgroup reads in pass p + 1 are those it wrote itself in pass p (same tiles, same CU, same L2); its
// L1 is invalidated after every wait (acquire).  Rows and tickets were write-through / atomic already.  Every wait is
// bounded (spin_limit polls): if the grid is not co-resident after all - another process took CUs - the waiting
// workgroups leave; the control block is only ever advanced by a complete pass, so what the host finds at its next
// poll is a consistent state and the per-pass path goes on from it.
// Bit-identical to per-pass launches: same geometry, same sums in the same order, same decide code.
// Registers: the chain needs 240 VGPRs per launch and, inside the pass loop, 274 - one wave per SIMD - although nothing
// of a pass outlives it but a few addresses.  amdgpu_waves_per_eu(2, 2) holds the allocator to 256: it then parks ~25
// pass-loop invariants in scratch, stored once per launch and loaded once per PASS, outside the tile loops
// (tests/test_abi.py checks exactly that: a bounded private segment, no scratch instruction at loop depth >= 2).
// What the workgroups need of the control block to run the next pass: THREE words in one cache line, published by
// the deciding wave; word 0 is written last and doubles as the barrier.  (Measured with the phase stamps of
// tools/persist_phases.py: 489 workgroups fetching the 52 words of the block itself behind every barrier - 25 000
// loads of four cache lines, all served by one memory channel - took 12 us per pass; one descriptor line per group of
// workgroups took as long, because 62 scattered write-through stores leave the deciding wave one after the other.)
//   [0] pass_seq | go << 32 | cur << 40 | prev << 44 | ring << 48     [1] lr     [2] nit
// The momentum factor of the next trial is beta_ring[nit % ZF_RING] (zf_resolve_beta with nothing lagging).
constexpr int ZF_PDESC_WORDS = 3;
template <bool NESTEROV, bool BOX, bool NT>
__global__ __launch_bounds__(ZF_BLOCK) __attribute__((amdgpu_waves_per_eu(2, 2))) void zf_persist_kernel(zf_step_args A, int npass, unsigned spin_limit) {
    constexpr int S = ZF_MAX_SUB;
    constexpr int CW = (int)(sizeof(zf_control) / 8);
    static_assert(sizeof(zf_control) % 8 == 0 && CW <= 64, "one lane per word of the control block");
    static_assert(S == 16, "the persistent kernel holds the 16-trial full chain");
    __shared__ double lds[ZF_WAVES * S * ZF_NPART];
    __shared__ zf_d2 stage[ZF_GLDS_NST * ZF_GLDS_STAGE_UNITS];
    __shared__ zf_control s_ctl;                      // the deciding workgroup's copy of the control block
    __shared__ unsigned long long s_desc[ZF_PDESC_WORDS];
    __shared__ int s_go;
    unsigned long long* gw = reinterpret_cast<unsigned long long*>(A.ctl_rw);
    unsigned long long* lw = reinterpret_cast<unsigned long long*>(&s_ctl);
    unsigned long long* my_desc = A.pdesc;
    if (threadIdx.x == 0) s_ctl.pass_seq = 0;
    // the first pass reads the block as every per-pass kernel does (written by an earlier launch)
    zf_pass_head HD = zf_head_of(A.ctl);
    if (A.ctl->status != ZF_RUNNING || A.ctl->lag != 0 || zf_fresh_len(A.ctl) != S) return;
#pragma unroll 1
    for (int p = 0; p < npass; ++p) {
        const int seq = A.pass_seq + p;
#ifdef ZF_PERSIST_DEBUG   // (timestamps of the phases of every pass and workgroup: tools/persist_phases.py)
        long long* dbg = reinterpret_cast<long long*>(A.hist) + ((int64_t)p * gridDim.x + blockIdx.x) * 8;
        if (A.hist && threadIdx.x == 0) {
            dbg[0] = wall_clock64();
            dbg[6] = gridDim.x;
        }
#endif
        if (A.pass_log && blockIdx.x == 0 && threadIdx.x == 0) A.pass_log[A.pass_slot] = A.pass_tag | zf_log_shape(0, S, p + 1);
        const double v = zf_trial_body<true, NESTEROV, BOX, NT, S, 0, false, S, true>(A, lds, HD, 0, S, stage);
#ifdef ZF_PERSIST_DEBUG
        if (A.hist && threadIdx.x == 0) dbg[2] = wall_clock64();
#endif
        zf_step_args T = A;           // this pass: decided on the deciding workgroup's own copy of the block
        T.ctl = A.ctl_rw;
        T.ctl_rw = &s_ctl;
        T.pass_seq = seq;
        T.decide = 1;
        zf_pass_tail<S, true, true>(T, v);
        __syncthreads();
#ifdef ZF_PERSIST_DEBUG
        if (A.hist && threadIdx.x == 0) {
            dbg[3] = wall_clock64();
            dbg[5] = (s_ctl.pass_seq == seq) ? 1 : 0;
        }
#endif
        if (s_ctl.pass_seq == seq) {
            // this workgroup decided the pass (zf_pass_tail left its number in the copy): publish the block for the
            // host and for later launches, then one descriptor per group, its word 0 last
            if (threadIdx.x < 64) {
                const int lane = threadIdx.x;
                if (lane < CW) __hip_atomic_store(gw + lane, lw[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int go = (s_ctl.status == ZF_RUNNING && s_ctl.lag == 0 && zf_fresh_len(&s_ctl) == S) ? 1 : 0;
                unsigned long long w[ZF_PDESC_WORDS];
                w[0] = (unsigned long long)(unsigned)seq | ((unsigned long long)go << 32) | ((unsigned long long)(s_ctl.cur & 15) << 40) |
                       ((unsigned long long)(s_ctl.prev & 15) << 44) | ((unsigned long long)(s_ctl.ring_size & 15) << 48);
                w[1] = (unsigned long long)__double_as_longlong(s_ctl.lr);
                w[2] = (unsigned long long)s_ctl.nit;
                if (lane >= 1 && lane < ZF_PDESC_WORDS) __hip_atomic_store(A.pdesc + lane, w[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0) {
                    __hip_atomic_store(A.pdesc, w[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                    for (int k = 0; k < ZF_PDESC_WORDS; ++k) s_desc[k] = w[k];
                    s_go = 1;
                }
            }
        } else if (threadIdx.x == 0) {
            int ok = 0;
            for (unsigned k = 0; k < spin_limit; ++k) {
                const unsigned long long w0 = __hip_atomic_load(my_desc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((int)(unsigned)w0 == seq) {
                    s_desc[0] = w0;
                    ok = 1;
                    break;
                }
                __builtin_amdgcn_s_sleep(4);
            }
            if (ok) {
#pragma unroll
                for (int k = 1; k < ZF_PDESC_WORDS; ++k) s_desc[k] = __hip_atomic_load(my_desc + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            s_go = ok;
        }
        __syncthreads();
#ifdef ZF_PERSIST_DEBUG
        if (A.hist && threadIdx.x == 0) dbg[4] = wall_clock64();
#endif
        if (!s_go || ((s_desc[0] >> 32) & 1) == 0) return;   // gave up waiting / the next pass is not a full chain
        // No agent-scope acquire here: its buffer_inv sc1 walks the L2, and 2000 waves issuing one behind every barrier
        // cost 11 us per pass (phase stamps).  What this workgroup reads next and somebody else wrote - rows, tickets,
        // the descriptor - is read past the caches anyway; the iterates it reads are those it stored itself, through
        // the L1 it reads them from.  The L1 of this CU alone is dropped all the same (sc0: a CU-local operation).
#ifndef ZF_PERSIST_NO_INV
        asm volatile("buffer_inv sc0" ::: "memory");
#endif
        // (wave-uniform values out of LDS into scalar registers: left in vector registers, lr, nit and the sixteen
        //  momentum factors loaded through them cost the chain its second wave per SIMD)
        const unsigned cp = __builtin_amdgcn_readfirstlane((unsigned)(s_desc[0] >> 40));
        HD.cur = (int)(cp & 15);
        HD.prev = (int)((cp >> 4) & 15);
        HD.ring = (int)((cp >> 8) & 15);
        HD.lr = zf_uniform_f64(__longlong_as_double((long long)s_desc[1]));
        HD.nit = (int64_t)zf_uniform_u64(s_desc[2]);
        HD.beta_next = NESTEROV ? A.beta_ring[HD.nit % ZF_RING] : 0.0;
    }
}

