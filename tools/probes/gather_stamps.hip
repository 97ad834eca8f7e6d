// Where does ONE unit of the f32 gather kernel spend its time?  The production kernel (csrc/conv3d.hip compiled into this
// program with -DMVX_GATHER_STAMPS) stamps s_memtime after every barrier of one unit's K stages; this program runs a layer
// shape, reads the stamps back and prints the time of each phase of a stage:
//   [top barrier] store halo + weight row 0 | [barrier] row 0 MFMAs | store row 1 | row 1 MFMAs | store row 2 | row 2 MFMAs
// build: hipcc --offload-arch=gfx950 -O3 -DMVX_GATHER_STAMPS -I include -I mvxnet-makise_amd/csrc tools/probes/gather_stamps.hip -o gather_stamps.bin
// usage: gather_stamps.bin <frames> <h> <w> <cin> <cout> [tile of the stamped unit / strip] [depth planes of a 3-D layer] [1 = the weight-gradient kernel]
#include "conv3d.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>

void mvxi_count_launch() {}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char **argv) {
    const int F = argc > 1 ? atoi(argv[1]) : 1, h = argc > 2 ? atoi(argv[2]) : 44, w = argc > 3 ? atoi(argv[3]) : 50;
    const int cin = argc > 4 ? atoi(argv[4]) : 256, cout = argc > 5 ? atoi(argv[5]) : 256;
    const int tile = argc > 6 ? atoi(argv[6]) : 5;
    const int planes = argc > 7 ? atoi(argv[7]) : 0;        // > 0: a 3-D layer (one frame of `planes` depth planes, stride 1) like the CML's conv2
    const int P = planes > 0 ? planes : 1;
    const size_t nx = (size_t)F * P * h * w * cin, ny = (size_t)F * P * h * w * cout, nw = (size_t)27 * cin * cout;
    std::vector<float> hx(nx), hw(nw);
    unsigned sd = 12345u;
    auto rnd = [&]() { sd = sd * 1664525u + 1013904223u; return ((sd >> 8) & 0xffff) / 65536.f - 0.5f; };
    for (auto &v : hx) v = rnd();
    for (auto &v : hw) v = rnd() * 0.05f;
    float *x, *wpk, *y, *bias;
    unsigned *counter;
    CK(hipMalloc(&x, nx * 4)); CK(hipMalloc(&wpk, nw * 4)); CK(hipMalloc(&y, ny * 4)); CK(hipMalloc(&bias, cout * 4));
    CK(hipMalloc(&counter, 64));
    CK(hipMemcpy(x, hx.data(), nx * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(wpk, hw.data(), nw * 4, hipMemcpyHostToDevice));
    CK(hipMemset(bias, 0, cout * 4));
    int unit[3] = {tile, planes > 0 ? planes / 2 : 0, 0};
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_unit), unit, sizeof(unit)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms = 0.f;
    for (int it = 0; it < 5; ++it) {
        CK(hipMemset(counter, 0, 64));
        CK(hipEventRecord(e0, 0));
        int rc = planes > 0 ? mvx_conv3d_forward(x, wpk, bias, y, nullptr, planes, planes, h, w, cin, cout, 1, 1, MVX_FLAG_RELU, counter, nullptr)
                            : mvx_conv2d_forward_frames(x, wpk, bias, y, nullptr, h, w, cin, cout, MVX_FLAG_RELU, nullptr, 1e-5, nullptr, counter, F, nullptr);
        if (rc) { fprintf(stderr, "forward: %d\n", rc); return 1; }
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
    }
    unsigned long long st[256];
    if (argc > 8 && atoi(argv[8]) == 1) {
        // ---- weight-gradient kernel (conv3d_wgrad4) of the same 2-D layer: stamps of workgroup (strip = tile argument, kd 1, chunk 0, block 0)
        float *dz, *dw;
        void *ws;
        CK(hipMalloc(&dz, ny * 4)); CK(hipMalloc(&dw, (size_t)cout * cin * 9 * 4));
        CK(hipMemcpy(dz, hx.data(), (ny < nx ? ny : nx) * 4, hipMemcpyHostToDevice));
        const size_t wsb = mvx_conv2d_wgrad_workspace_bytes_frames(h, w, cin, cout, F);
        CK(hipMalloc(&ws, wsb));
        int wunit[3] = {tile, 1 * (cin / 64) + 0, 0};
        CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_unit), wunit, sizeof(wunit)));
        for (int it = 0; it < 5; ++it) {
            CK(hipEventRecord(e0, 0));
            int rc = mvx_conv2d_wgrad_frames(x, dz, dw, h, w, cin, cout, 0, ws, wsb, F, nullptr);
            if (rc) { fprintf(stderr, "wgrad: %d\n", rc); return 1; }
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
        }
        CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(g_stamps), sizeof(st)));
        const int nsteps = (int)st[251];
        printf("wgrad4 %d frames %dx%d %d->%d: three launches %.1f us; workgroup strip %d: %d steps (tiles)\n", F, h, w, cin, cout, ms * 1e3, tile, nsteps);
        printf("%-6s %12s %12s %12s\n", "step", "to barrier", "LDS stores", "MFMA issue");
        double sm[3] = {0, 0, 0};
        for (int i = 0; i < nsteps && 3 + 3 * i < 250; ++i) {
            const double a = (double)(st[1 + 3 * i] - (i == 0 ? st[0] : st[3 * i])), b = (double)(st[2 + 3 * i] - st[1 + 3 * i]),
                         c = (double)(st[3 + 3 * i] - st[2 + 3 * i]);
            printf("%-6d %12.0f %12.0f %12.0f\n", i, a, b, c);
            sm[0] += a; sm[1] += b; sm[2] += c;
        }
        printf("sum    %12.0f %12.0f %12.0f   loop %.0f clocks (5 or 4 taps x 64 k-steps x 64 clocks = 20,480 / 16,384 of MFMA per wave and step, two waves per SIMD);"
               " slab stores %.0f\n", sm[0], sm[1], sm[2], (double)(st[250] - st[0]), (double)(st[252] - st[250]));
        return 0;
    }
    CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(g_stamps), sizeof(st)));
    const int nst = (planes > 0 ? 3 : 1) * (cin / 32);          // depth taps x chunks
    printf("layer %d frames %dx%d %d->%d: launch %.1f us; unit tile %d; s_memtime ticks (shader clocks)\n", F, h, w, cin, cout, ms * 1e3, tile);
    printf("%-6s %10s %10s %10s %10s %10s %10s %10s\n", "stage", "top-wait", "st h+w0", "mfma r0", "st w1", "mfma r1", "st w2", "mfma r2");
    double sum[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int s = 0; s < nst; ++s) {
        const unsigned long long *b = st + 1 + 6 * s;
        const unsigned long long prev = s == 0 ? st[0] : st[6 * s];
        double d[7] = {(double)(b[0] - prev), (double)(b[1] - b[0]), (double)(b[2] - b[1]), (double)(b[3] - b[2]), (double)(b[4] - b[3]),
                       (double)(b[5] - b[4]), (double)(st[1 + 6 * (s + 1)] - b[5])};
        // "top-wait" of stage s = time from the previous stamp (row 2 MFMAs start of the stage before, or unit start) to the top barrier;
        // the last column = row-2 MFMAs issue + the next top barrier
        printf("%-6d", s);
        for (int k = 0; k < 7; ++k) { printf(" %10.0f", d[k]); sum[k] += d[k]; }
        printf("\n");
    }
    printf("%-6s", "sum");
    for (int k = 0; k < 7; ++k) printf(" %10.0f", sum[k]);
    printf("\nK loop %.0f ticks; epilogue (bias, stores, statistics) %.0f; to the next unit's K loop entry (work queue) %.0f; its first operands in LDS %.0f\n",
           (double)(st[1 + 6 * nst] - st[0]), (double)(st[200] - st[1 + 6 * nst]), (double)(st[201] - st[200]), (double)(st[202] - st[201]));
    return 0;
}
