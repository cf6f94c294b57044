// experiments.h -- the measurement hooks of the SEQ / ROW drivers, in ONE place.
//
// A product build (the Makefile) never defines PQ_EXPERIMENTS: every hook below is then the plain operation and the constants are the
// shipped values.  `scripts/ab_build.sh <name> "-DPQ_EXPERIMENTS -DPQ_EXP_..."` builds an A/B variant of the library (ab/libpq_<name>.so,
// selected at run time with PQ_LIB_PATH); the switches and what they answered are listed in DESIGN.md section 4:
//   PQ_EXP_NOSTORE       every output store replaced by a never-true one: how much of a step is the output traffic?
//   PQ_EXP_PLAIN_STORES  plain instead of non-temporal stores
//   PQ_EXP_NOLOAD        tile loads replaced by constants (with NOCOMPUTE: what do the input reads cost?)
//   PQ_EXP_NOCOMPUTE     outputs = the first input: the traffic and the hand-off machinery alone
//   PQ_EXP_STOREONLY     the store replica: the storer waves of the tiled job bodies issue exactly the step's stores (grids, addresses, piece
//                        sizes, non-temporal policy) from register values; the compute wave returns at once, no LDS traffic, no barriers
//   PQ_PROFILE_WAVES     s_memtime accounting of the compute wave per job kind (load wait + LDS fill, rows, hand-off), SIMD histogram
//   PQ_STORER_ACC=2|4    the storer wave keeps 2 / 4 out tiles and stores them back to back
//   PQ_PF2_MAX=<n>       a second tile of register prefetch for ops whose inputs need <= n VGPRs
//   PQ_SEQ_NUM_VGPR=<n>  register cap (in PAIRS: 96 = 192 VGPRs) on every stand-alone op kernel: which ops spill under the job kernel's cap?
//   PQ_ANALYZE_LIGHT=X(Op)...  (suite.hip) the light job kernel with a subset of its ops: per-op registers / spills INSIDE the job kernel
//                        (scripts/kernel_resources.sh; such a build is never linked)
#pragma once

#if defined(PQ_EXPERIMENTS) && defined(PQ_EXP_NOSTORE)
#define PQ_EXP_NOSTORE_ON 1
#else
#define PQ_EXP_NOSTORE_ON 0
#endif

#if defined(PQ_EXPERIMENTS) && defined(PQ_EXP_STOREONLY)
#define PQ_EXP_STOREONLY_ON 1
#else
#define PQ_EXP_STOREONLY_ON 0
#endif
#if defined(PQ_EXPERIMENTS) && defined(PQ_EXP_SO_ALLPAIR) // with PQ_EXP_STOREONLY: every column of a pair-mode job in 128-byte pieces, whatever the register cap
#define PQ_EXP_SO_ALLPAIR_ON 1
#else
#define PQ_EXP_SO_ALLPAIR_ON 0
#endif
// (SO: a constant of run_seq_lds -- the replica applies to this op)
#define PQ_HOOK_STORER_BARRIER() do { if constexpr (!SO) __builtin_amdgcn_s_barrier(); } while (0)
#define PQ_HOOK_STORER_PULL(q, kk, i) (SO ? make_double2((double)(kk), (double)((i) + lane)) : make_double2((q)[0], (q)[1]))
#define PQ_HOOK_STORER_PULL_G(expr, kk, i) (SO ? make_double2((double)(kk), (double)((i) + lane)) : (expr))

#ifndef PQ_EXPERIMENTS
// ---------------------------------------------------------------- product: the plain operations
#define PQ_HOOK_STORE2(w, ptr) __builtin_nontemporal_store((w), (ptr))
#define PQ_HOOK_ROW_STORE(v, ptr) __builtin_nontemporal_store((v), (ptr))
#define PQ_HOOK_TILE_LOAD(src, t0, i) (*reinterpret_cast<const double2 *>(src))
#define PQ_HOOK_FAST_OK(cond) (cond)
#define PQ_HOOK_FAST_ROWS(Op, FU, NOUT, op, t, xs, ys) fast_rows<Op, FU>(op, t, xs, ys)
#define PQ_PROF_T(v)
#define PQ_PROF_ADD(k, dt)
#define PQ_PROF_SIMD(wave)
#define PQ_STORER_ACC 1
#define PQ_PF2_MAX 0
#define PQ_HOOK_SEQ_KERNEL_ATTR
#else
// ---------------------------------------------------------------- scripts/ab_build.sh only
#if defined(PQ_EXP_NOSTORE)
#define PQ_HOOK_STORE2(w, ptr) do { if ((w).x == 1.2345e-300) *(ptr) = (w); } while (0)
#define PQ_HOOK_ROW_STORE(v, ptr) do { if ((double)(v) == 123456789.0) *(ptr) = (v); } while (0)
#elif defined(PQ_EXP_PLAIN_STORES)
#define PQ_HOOK_STORE2(w, ptr) (*(ptr) = (w))
#define PQ_HOOK_ROW_STORE(v, ptr) (*(ptr) = (v))
#else
#define PQ_HOOK_STORE2(w, ptr) __builtin_nontemporal_store((w), (ptr))
#define PQ_HOOK_ROW_STORE(v, ptr) __builtin_nontemporal_store((v), (ptr))
#endif
#ifdef PQ_EXP_NOLOAD
#define PQ_HOOK_TILE_LOAD(src, t0, i) make_double2((double)(t0), (double)(i))
#else
#define PQ_HOOK_TILE_LOAD(src, t0, i) (*reinterpret_cast<const double2 *>(src))
#endif
#ifdef PQ_EXP_NOCOMPUTE
#define PQ_HOOK_FAST_OK(cond) (true)
#define PQ_HOOK_FAST_ROWS(Op, FU, NOUT, op, t, xs, ys) do { for (int u__ = 0; u__ < FU; u__++) for (int k__ = 0; k__ < NOUT; k__++) ys[u__][k__] = xs[u__][0]; } while (0)
#else
#define PQ_HOOK_FAST_OK(cond) (cond)
#define PQ_HOOK_FAST_ROWS(Op, FU, NOUT, op, t, xs, ys) fast_rows<Op, FU>(op, t, xs, ys)
#endif
#ifdef PQ_PROFILE_WAVES // [job kind][load wait + LDS fill, rows, hand-off, tiles]; rows 120..123: SIMD histogram per wave role
static __device__ unsigned long long pq_prof[128][4];
template <class Op, class = void> struct ProfId { static constexpr int value = 0; };
template <class Op> struct ProfId<Op, decltype((void)Op::SEQ_ID)> { static constexpr int value = Op::SEQ_ID & 127; };
#define PQ_PROF_T(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#define PQ_PROF_ADD(k, dt) do { if (lane == 0) atomicAdd(&pq_prof[ProfId<Op>::value][k], (unsigned long long)(dt)); } while (0)
#define PQ_PROF_SIMD(wave) do { if (lane == 0) atomicAdd(&pq_prof[120 + (__builtin_amdgcn_s_getreg(2308) & 3)][wave], 1ULL); } while (0)
#else
#define PQ_PROF_T(v)
#define PQ_PROF_ADD(k, dt)
#define PQ_PROF_SIMD(wave)
#endif
#ifndef PQ_STORER_ACC
#define PQ_STORER_ACC 1
#endif
#ifndef PQ_PF2_MAX
#define PQ_PF2_MAX 0
#endif
#ifdef PQ_SEQ_NUM_VGPR
#define PQ_HOOK_SEQ_KERNEL_ATTR __attribute__((amdgpu_num_vgpr(PQ_SEQ_NUM_VGPR)))
#else
#define PQ_HOOK_SEQ_KERNEL_ATTR
#endif
#endif
