// gather_cabi.cpp -- the per-step exchange of the symbol-sharded backtest through the product's own C ABI, WITHOUT torch or Python:
// a 625-symbol x 2 520-day MACD-cross backtest step (pq_backtest_macd_cross) with
//   kernel_only  no exchange
//   serial       pq_gather_summaries behind every step, on the step's stream
//   overlapped   pq_gather_summaries_begin / _end, double-buffered on the communicator's own stream
// at a world of one (RCCL's one-rank all-gather).  scripts/bench_gather.py measures the same three through Python + torch streams; the
// difference between the two is the harness, what is left is the runtime + RCCL (scripts/ubench/xstream.hip: the runtime alone).
// build (on the GPU box):  hipcc -O2 -std=c++17 gather_cabi.cpp -o gather_cabi -I../../include -L../../polars_quant_amd -lpolars_quant_hip -Wl,-rpath,$PWD/../../polars_quant_amd
#include <hip/hip_runtime.h>
#include <chrono>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "pq_hip.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#define PQ(x) do { pq_status s_ = (x); if (s_ != PQ_OK) { fprintf(stderr, "%s: status %d: %s (line %d)\n", #x, (int)s_, pq_last_error(), __LINE__); exit(1); } } while (0)
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 625, T = 2520, pitch = 2528;
    const int steps = argc > 2 ? atoi(argv[2]) : 400;
    pq_ctx *ctx = nullptr;
    PQ(pq_ctx_create(0, nullptr, &ctx));
    std::vector<double> h((size_t)n * pitch, 0.0);
    unsigned long long rng = 0x5EED0002ULL;
    for (int64_t s = 0; s < n; s++) {
        double p = 50.0 + (double)(s % 97);
        for (int64_t t = 0; t < T; t++) {
            rng = rng * 6364136223846793005ULL + 1442695040888963407ULL;
            p *= 1.0 + 0.02 * ((double)(rng >> 11) / 9007199254740992.0 - 0.5);
            h[(size_t)s * pitch + t] = p;
        }
    }
    double *close, *cur[3], *local[2], *all[2];
    CK(hipMalloc(&close, h.size() * 8));
    CK(hipMemcpy(close, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    for (auto &c : cur) CK(hipMalloc(&c, h.size() * 8));
    for (int k = 0; k < 2; k++) { CK(hipMalloc(&local[k], (size_t)n * 8 * 8)); CK(hipMalloc(&all[k], (size_t)n * 8 * 8)); }
    unsigned char id[PQ_COMM_ID_BYTES];
    PQ(pq_comm_unique_id(id));
    PQ(pq_comm_init(ctx, 0, 1, id));
    pq_batch b{};
    b.n_series = n; b.len = T; b.stride = pitch; b.offsets = nullptr;
    pq_bt_params prm{};
    prm.initial_capital = 100000.0; prm.buy_slippage = 0.0; prm.sell_slippage = 0.0; prm.buy_commission_rate = 0.0003; prm.sell_commission_rate = 0.0003;
    prm.min_commission = 5.0; prm.position_size = 1.0;
    auto step = [&](int slot) { PQ(pq_backtest_macd_cross(ctx, &b, close, 12, 26, 9, &prm, cur[0], cur[1], cur[2], local[slot])); };
    auto run = [&](int mode) {
        double best = 1e30;
        for (int rep = 0; rep < 3; rep++) {
            PQ(pq_ctx_sync(ctx)); PQ(pq_comm_sync(ctx));
            const double t0 = now_us();
            for (int k = 0; k < steps; k++) {
                const int slot = k & 1;
                if (mode == 2) PQ(pq_gather_summaries_end(ctx, slot)); // the exchange of step k - 2 used this slot
                step(slot);
                if (mode == 1) PQ(pq_gather_summaries(ctx, local[slot], n, all[slot]));
                if (mode == 2) PQ(pq_gather_summaries_begin(ctx, local[slot], n, all[slot], slot));
            }
            if (mode == 2) { PQ(pq_gather_summaries_end(ctx, 0)); PQ(pq_gather_summaries_end(ctx, 1)); }
            PQ(pq_ctx_sync(ctx)); PQ(pq_comm_sync(ctx));
            const double dt = (now_us() - t0) / steps;
            best = dt < best ? dt : best;
        }
        return best;
    };
    // host cost of the calls alone (enqueue without waiting): how close is the HOST to being the bottleneck of a 50 us step?
    auto host_cost = [&](int mode) {
        PQ(pq_ctx_sync(ctx)); PQ(pq_comm_sync(ctx));
        const int m = 50;
        const double t0 = now_us();
        for (int k = 0; k < m; k++) {
            const int slot = k & 1;
            if (mode == 2) PQ(pq_gather_summaries_end(ctx, slot));
            step(slot);
            if (mode == 1) PQ(pq_gather_summaries(ctx, local[slot], n, all[slot]));
            if (mode == 2) PQ(pq_gather_summaries_begin(ctx, local[slot], n, all[slot], slot));
        }
        const double dt = (now_us() - t0) / m;
        if (mode == 2) { PQ(pq_gather_summaries_end(ctx, 0)); PQ(pq_gather_summaries_end(ctx, 1)); }
        PQ(pq_ctx_sync(ctx)); PQ(pq_comm_sync(ctx));
        return dt;
    };
    run(0); run(1); run(2);
    const double k0 = run(0), k1 = run(1), k2 = run(2);
    printf("{\"symbols\": %lld, \"days\": %lld, \"world\": 1, \"steps\": %d, \"harness\": \"C, no torch\",\n \"us_per_step\": {\"kernel_only\": %.2f, \"serial\": %.2f, \"overlapped\": %.2f},\n"
           " \"host_enqueue_us_per_step\": {\"kernel_only\": %.2f, \"serial\": %.2f, \"overlapped\": %.2f}}\n",
           (long long)n, (long long)T, steps, k0, k1, k2, host_cost(0), host_cost(1), host_cost(2));
    PQ(pq_comm_destroy(ctx));
    PQ(pq_ctx_destroy(ctx));
    return 0;
}
