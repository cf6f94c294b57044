// suite_jobs.h -- the device-visible job record of the job grids (suite.hip), the list of recordable ops, the register cap.
#pragma once
#include "ops_backtest.h"
#include "ops_fused.h"

struct SeqJob { // device-visible
    int kind, nin, nout, cost; // cost: estimated solo duration in microseconds (SeqTraits)
    int heavy, masked;
    int cls;                   // CLS_*: the grid the job runs in (suite_finalize)
    int prio;                  // s_setprio of the job's waves: by its cost relative to the longest job of the phase (suite_finalize)
    int unal;                  // rows only 8-byte aligned: the grid of its class runs the 8-byte form of the tiled body (seq_jobs_kernel<3>)
    double summary_bytes;
    int alg_cols;                  // f64 column transfers credited (per reference call)
    unsigned lds_bytes, tile_bytes; // lds_bytes 0 = run the gather body
    int tile_k;                     // rows per tile of the tiled body (SeqTile<Op>::K)
    int ts_len, ts_skip, ts_row0;   // time-split job (pq_dev.h TsOk): the rows of its range (0: the batch's len), the leading tiles it keeps to itself,
                                    // its first row (the job's columns are in / out / the op's own pointers + ts_row0: the recorded pointers stay the
                                    // columns' bases, which is what the hazard tracking keys on)
    float alg_frac;                 // share of the op's algorithmic bytes credited to this job (1 unless time-split)
    const double *in[6];
    double *out[8];
    const void *xr[4]; // columns the job touches outside its tile path (signal / benchmark columns read; a summary table, derived columns
    void *xw[4];       // written): what the recording's hazard tracking and the small-shard schedule's dependencies need to know
    alignas(8) unsigned char op[1024];
};
static_assert(sizeof(BtArgs) <= 1024, "BtArgs must fit a job slot");

// every recordable SEQ op: X(Type).  Kernel variants: the light job kernel is capped at 192 VGPRs (PQ_NV0 below; every op of the list
// fits without scratch); an op marked HEAVY (none at present: STOCH and the Hilbert pipeline were slimmed in round 3) and the lane-form
// backtest scan run in a second kernel with the full 256.
#define SEQ_OPS_LIGHT(X)                                                                                             \
    X(SmaOp) X(EmaOp) X(BbandsOp) X(DemaOp) X(TemaOp) X(T3Op) X(WmaOp) X(KamaOp) X(MidpointOp) X(MidpriceOp) X(SarextOp) \
    X(MavpPickOp) X(MavpSelOp<SmaOp>) X(MavpSelOp<EmaOp>) X(MavpSelOp<WmaOp>) X(MavpSelOp<DemaOp>) X(MavpSelOp<TemaOp>)  \
    X(MavpSelOp<T3Op>) X(MavpSelOp<KamaOp>)                                                                          \
    X(CmoOp) X(RsiOp) X(MacdOp) X(TrixOp) X(UltoscOp) X(MfiOp) X(DmOp<0>) X(DmOp<1>) X(DmOp<2>) X(DmRawOp<true>)      \
    X(DmRawOp<false>) X(SmaTpOp)                                                                                     \
    X(TrimaOp) X(MaDiffOp<0>) X(MaDiffOp<1>) X(MacdextOp) X(StochOp<0>) X(StochOp<1>) X(StochAllOp) X(StochRsiOp) X(CciOp)       \
    X(DmAllOp<true>) X(DmAllOp<false>) X(MavpBlockOp<1>) X(MavpSma16Op) X(MavpSma8Op) X(MavpSma32Op) X(UltoscOp8)                                          \
    X(AtrOp<false>) X(AtrOp<true>) X(ObvOp) X(AdOp<false>) X(AdOp<true>) X(HtOp<0>) X(HtOp<1>) X(HtOp<2>) X(HtOp<3>) X(HtOp<4>) X(HtAllOp) X(HtAll6Op) X(BtMacdOp) X(LevOp)                                     \
    X(EmaAllOp) X(AtrAllOp) X(DmPairOp) X(AdAllOp) X(MacdPairOp) X(ApoPpoOp) X(SarPairOp) X(VolumeAllOp) X(DmiAtrOp) X(CmoRsiOp) X(SmaDupOp)
#if defined(PQ_EXPERIMENTS) && defined(PQ_ANALYZE_LIGHT) // analysis builds (never linked): the light job kernel with a subset of its ops (experiments.h)
#undef SEQ_OPS_LIGHT
#define SEQ_OPS_LIGHT(X) PQ_ANALYZE_LIGHT
#endif
#define SEQ_OPS_HEAVY(X) // (none since the Hilbert pipeline keeps its delay lines in LDS rings; the class and its kernel remain for ops marked HEAVY)
// V = 0: LDS bodies of the light ops (2 waves/SIMD, capped at 192 VGPRs: PQ_NV0 below), 1: LDS bodies of the heavy ops, 2: gather
// bodies of every op + the backtest scan (one wave per workgroup; the fallback for very long windows / unaligned columns).
#ifndef PQ_LB0
// waves per SIMD the light kernel is compiled for: 2.  At 3 (168 VGPRs, round 2) the three widest jobs (EMA x 4, the volume family, the
// DM system) spilled 359 registers / 272 B of scratch per lane, and LDS holds a CU to four of the LONG grid's workgroups = 2 waves /
// SIMD anyway.  The actual cap is PQ_NV0 (192).  Table: profiles/r03_kernel_resources.txt, csrc/suite.resources.txt (every build)
#define PQ_LB0 2
#endif
// Register cap of the light kernel: 192 VGPRs (`amdgpu_num_vgpr` counts register PAIRS on gfx950: 96).  Two job waves then leave 128
// of a SIMD's 512 registers free, which is what one wave of the pattern kernel (118) or of the wave-per-symbol backtest (122) needs:
// at the uncapped 199 (200 allocated) neither fits beside two job waves and the pattern kernel -- the last chain of a step to finish --
// only advances where a job workgroup has retired.  A/B in one session: 4.60 -> 4.53 ms per step.  PQ_NV0=0: no cap.
// The gather kernel (seq_jobs_kernel<2>: the per-lane bodies of every op in one switch -- ragged batches inside a recording, windows
// beyond 64 KB of LDS) is compiled for ONE wave per SIMD: its workgroups are single waves, the widest bodies (the Hilbert family's register
// delay lines, MAVP's candidate sums) need ~300 registers, and under the 256 of two waves per SIMD the kernel spilled 90 of them to
// scratch (176 B per lane) for EVERY op of the switch.  A per-lane walk is bound by its L1 accesses, not by occupancy.
#ifndef PQ_LB2
#define PQ_LB2 1
#endif
#ifndef PQ_NV0
#define PQ_NV0 96
#endif
