// Microbenchmark behind DESIGN.md 4.2: at what rate can ONE CU pull a weight image that every CU of the chip reads, flow after flow?
// Models the row-split flow-step kernels (coupling_fused_rs_kernel): one workgroup per CU, NW waves, every wave streams its 1/NW of a
// `bytes`-sized image per flow through a D-deep register ring of 1-KiB wave-loads (global_load_dwordx4), `nflows` images per launch.
//   mode 0  distinct image per flow (as in a level launch: 16 x 1.25 MB does not stay in a 4 MiB L2)
//   mode 1  every flow reads image 0 (L2-resident after the first pass)
//   mode 2  mode 0 + L2 prefetch: while flow f streams, the workgroup touches its 1/32 share of image f+AHEAD with one dword per 128-B
//           line (one wave-instruction = 64 lines = 8 KiB), so that the XCD's L2 holds the image before the demand stream arrives
//   mode 3  mode 0, every workgroup starts its sweep at a different offset (rotated walk: a CU leads on 1/32 of the image only)
// build: hipcc -O3 --offload-arch=gfx950 tools/l2_stream_probe.hip -o gpurun_out/l2_stream_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int NW, int D>
__global__ void __launch_bounds__(64 * NW) probe(const char* __restrict__ base, long bytes, int nflows, int mode, int ahead, unsigned* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long per_wave = bytes / NW;                  // multiple of 1 KiB
    const int nfrag = (int)(per_wave / 1024);
    const int k = blockIdx.x / 8;                      // position of this workgroup inside its XCD (round-robin dispatch)
    u32x4 acc = {0, 0, 0, 0};
    for (int f = 0; f < nflows; ++f) {
        const char* img = base + (mode == 1 ? 0 : (long)f * bytes);
        const char* mine = img + wave * per_wave + lane * 16;
        int rot = 0;
        if (mode == 3) rot = (int)(((long)k * nfrag / 32) % nfrag);
        if (mode == 2 && f + ahead < nflows) {
            // this workgroup's 1/32 of image f+ahead, one dword per 128-B line; NW waves x 64 lines x 128 B = NW x 8 KiB per round
            const char* nxt = base + (long)(f + ahead) * bytes;
            const long share = bytes / 32;
            for (long o = (long)wave * 8192 + lane * 128; o < share; o += NW * 8192) {
                unsigned v = *reinterpret_cast<const unsigned*>(nxt + (long)(k % 32) * share + o);
                acc[0] ^= v;
            }
        }
        u32x4 ring[D];
        const int nround = nfrag / D;                  // whole ring rounds (the remainder of the image is not read: `bytes_read` below)
#pragma unroll
        for (int i = 0; i < D; ++i) ring[i] = *reinterpret_cast<const u32x4*>(mine + (long)((i + rot) % nfrag) * 1024);
        for (int r = 0; r + 1 < nround; ++r) {         // steady state: consume fragment n, request fragment n + D
            const int nb = (r + 1) * D + rot;
#pragma unroll
            for (int i = 0; i < D; ++i) {
                acc ^= ring[i];
                int nn = nb + i;
                if (nn >= nfrag) nn -= nfrag;
                ring[i] = *reinterpret_cast<const u32x4*>(mine + (long)nn * 1024);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int i = 0; i < D; ++i) acc ^= ring[i];
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) out[blockIdx.x] = 1;
}

template <int NW, int D>
static void run(const char* buf, long bytes, int nflows, int nwg, unsigned* out) {
    for (int mode = 0; mode < 4; ++mode) {
        for (int ahead = 1; ahead <= (mode == 2 ? 2 : 1); ++ahead) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            std::vector<float> ms;
            for (int it = 0; it < 7; ++it) {
                hipEventRecord(e0, 0);
                hipLaunchKernelGGL((probe<NW, D>), dim3(nwg), dim3(64 * NW), 0, 0, buf, bytes, nflows, mode, ahead, out);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                float t;
                hipEventElapsedTime(&t, e0, e1);
                if (it >= 2) ms.push_back(t);
            }
            std::sort(ms.begin(), ms.end());
            const double t = ms[ms.size() / 2] * 1e-3;
            const double rd = (double)(bytes / NW / 1024 / D * D) * 1024 * NW;      // bytes a workgroup really reads per flow
            printf("NW=%d D=%2d nwg=%3d bytes=%7ld flows=%d mode=%d ahead=%d : %8.1f us/launch  %6.2f us/flow  %6.1f GB/s per CU  %5.2f TB/s chip\n", NW, D,
                   nwg, bytes, nflows, mode, ahead, t * 1e6, t * 1e6 / nflows, rd * (double)nflows / t / 1e9, rd * (double)nflows * nwg / t / 1e12);
            fflush(stdout);
        }
    }
}

int main(int argc, char** argv) {
    const int nflows = 16;
    const long maxbytes = 1310720;                     // 1.25 MiB: the 4x4 level's image
    char* buf;
    unsigned* out;
    hipMalloc(&buf, maxbytes * nflows);
    hipMalloc(&out, 4096);
    hipMemset(buf, 1, maxbytes * nflows);
    for (long bytes : {1310720L, 884736L}) {           // 4x4-level and 8x8-level images (rounded to NW x 1 KiB multiples)
        for (int nwg : {256, 32}) {
            run<4, 24>(buf, bytes, nflows, nwg, out);
            run<4, 12>(buf, bytes, nflows, nwg, out);
            run<8, 12>(buf, bytes, nflows, nwg, out);
            run<8, 24>(buf, bytes, nflows, nwg, out);
            run<16, 8>(buf, bytes, nflows, nwg, out);
        }
    }
    hipDeviceSynchronize();
    printf("done\n");
    return 0;
}
