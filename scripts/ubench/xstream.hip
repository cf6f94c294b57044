// xstream.hip -- what does a dependency between two HIP streams cost on this runtime?  Bare HIP, no torch.
// A "step" kernel of ~50 us on stream A and an "exchange" kernel of ~5 us on stream B, in the three shapes a per-step exchange can take:
//   serial   A: step, exchange, step, exchange ...                                   (one stream, no event)
//   pingpong A: step -> event -> B: exchange -> event -> A: next step                (B's result gates the next step)
//   oneway   A: step -> event -> B: exchange;  A goes on with the next step at once  (the double-buffered overlapped gather: B never gates A;
//                                                                                      the host polls hipEventQuery on B's event two steps later)
// each with timing-enabled and hipEventDisableTiming events, and with A / B created BEFORE or AFTER three idle side streams (the runtime
// maps streams onto its hardware queues in creation order).  Prints us per step; the cost of the dependency is the difference to `serial`
// (pingpong) or to the bare step (oneway).
// build: hipcc -O3 --offload-arch=gfx950 xstream.hip -o xstream
#include <hip/hip_runtime.h>
#include <chrono>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void spin(unsigned long long ticks, unsigned long long *sink) { // wall_clock64: 100 MHz
    const unsigned long long t0 = wall_clock64();
    unsigned long long t = t0;
    while (t - t0 < ticks) t = wall_clock64();
    if (sink && threadIdx.x == 0) *sink = t;
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Streams { hipStream_t a, b, idle[3]; };
static Streams make_streams(bool idle_first) {
    Streams s{};
    int lo = 0, hi = 0;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    if (idle_first) for (auto &q : s.idle) CK(hipStreamCreateWithPriority(&q, hipStreamNonBlocking, hi));
    CK(hipStreamCreateWithPriority(&s.a, hipStreamNonBlocking, hi));
    CK(hipStreamCreateWithPriority(&s.b, hipStreamNonBlocking, hi));
    if (!idle_first) for (auto &q : s.idle) CK(hipStreamCreateWithPriority(&q, hipStreamNonBlocking, hi));
    for (auto &q : s.idle) { spin<<<1, 64, 0, q>>>(100, nullptr); CK(hipStreamSynchronize(q)); } // every stream has been used once
    return s;
}
static void free_streams(Streams &s) {
    CK(hipStreamDestroy(s.a)); CK(hipStreamDestroy(s.b));
    for (auto &q : s.idle) CK(hipStreamDestroy(q));
}

enum Mode { SERIAL, PINGPONG, ONEWAY, STEP_ONLY };
static double run(const Streams &s, Mode m, unsigned flags, int n, unsigned long long step_ticks, unsigned long long xch_ticks, unsigned long long *sink) {
    const int NE = 8;
    hipEvent_t e1[NE], e2[NE];
    for (int i = 0; i < NE; i++) { CK(hipEventCreateWithFlags(&e1[i], flags)); CK(hipEventCreateWithFlags(&e2[i], flags)); }
    double best = 1e30;
    for (int rep = 0; rep < 3; rep++) {
        CK(hipDeviceSynchronize());
        const double t0 = now_us();
        for (int k = 0; k < n; k++) {
            const int i = k % NE;
            if (m == ONEWAY && k >= 2) { // the slot the exchange of step k - 2 used is about to be refilled: its exchange must be over
                const int j = (k - 2) % NE;
                if (hipEventQuery(e2[j]) == hipErrorNotReady) CK(hipEventSynchronize(e2[j]));
            }
            spin<<<1, 64, 0, s.a>>>(step_ticks, sink);
            if (m == SERIAL) spin<<<1, 64, 0, s.a>>>(xch_ticks, sink + 1);
            if (m == PINGPONG || m == ONEWAY) {
                CK(hipEventRecord(e1[i], s.a));
                CK(hipStreamWaitEvent(s.b, e1[i], 0));
                spin<<<1, 64, 0, s.b>>>(xch_ticks, sink + 1);
                CK(hipEventRecord(e2[i], s.b));
                if (m == PINGPONG) CK(hipStreamWaitEvent(s.a, e2[i], 0));
            }
        }
        CK(hipStreamSynchronize(s.a));
        CK(hipStreamSynchronize(s.b));
        const double dt = (now_us() - t0) / n;
        best = dt < best ? dt : best;
    }
    for (int i = 0; i < NE; i++) { CK(hipEventDestroy(e1[i])); CK(hipEventDestroy(e2[i])); }
    return best;
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 400;
    unsigned long long *sink;
    CK(hipMalloc(&sink, 64));
    const unsigned long long STEP = 5000, XCH = 500; // 50 us, 5 us
    printf("{\n \"step_us_nominal\": 50, \"exchange_us_nominal\": 5, \"steps_per_run\": %d, \"runs\": [\n", n);
    bool first = true;
    for (int idle_first = 0; idle_first < 2; idle_first++) {
        Streams s = make_streams(idle_first != 0);
        for (int timing = 0; timing < 2; timing++) {
            const unsigned flags = timing ? hipEventDefault : hipEventDisableTiming;
            const double step_only = run(s, STEP_ONLY, flags, n, STEP, XCH, sink);
            const double serial = run(s, SERIAL, flags, n, STEP, XCH, sink);
            const double ping = run(s, PINGPONG, flags, n, STEP, XCH, sink);
            const double oneway = run(s, ONEWAY, flags, n, STEP, XCH, sink);
            printf("%s  {\"streams_created\": \"%s\", \"events\": \"%s\", \"us_per_step\": {\"step_only\": %.2f, \"serial\": %.2f, \"pingpong\": %.2f, \"oneway\": %.2f},\n"
                   "   \"dependency_cost_us\": {\"pingpong_minus_serial\": %.2f, \"oneway_minus_step_only\": %.2f}}",
                   first ? "" : ",\n", idle_first ? "after three idle side streams" : "before three idle side streams", timing ? "timing" : "hipEventDisableTiming",
                   step_only, serial, ping, oneway, ping - serial, oneway - step_only);
            first = false;
        }
        free_streams(s);
    }
    printf("\n ]\n}\n");
    return 0;
}
