// What the matrix cores sustain under the board's power limit: MFMA-only loops on register-resident operands (no LDS, no
// memory traffic inside the loop), 8 waves per CU like the GEMM tile, for both bf16 shapes and for operand data of
// increasing switching activity.  Prints TFLOP/s with the shader clock and board power sampled from sysfs while each
// case runs for ~2 s.  A roofline priced at 2.4 GHz x 1017 flops / cycle / SIMD (2.5 PFLOP/s) is reachable only if the
// clock stays at 2.4 GHz; this tool measures the clock the board actually holds for this instruction mix and data.
//
//   hipcc -O3 --offload-arch=gfx950 -o tools/_build/mfma_power tools/mfma_power.hip && tools/_build/mfma_power
#include <hip/hip_runtime.h>

#include <atomic>
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dirent.h>
#include <string>
#include <thread>
#include <unistd.h>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define CK(x)                                                                                   \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } \
    } while (0)

// K = 1536 worth of MFMAs per accumulator between resets (as in the ViT-g GEMMs), `reps` resets per launch
template <int SHAPE>
__global__ __launch_bounds__(256) void k_mfma(const uint4* __restrict__ src, float* __restrict__ sink, int reps) {
    const int gid = blockIdx.x * 256 + threadIdx.x;
    if constexpr (SHAPE == 16) {
        bf16x8 a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            a[i] = __builtin_bit_cast(bf16x8, src[(size_t)gid * 8 + i]);
            b[i] = __builtin_bit_cast(bf16x8, src[(size_t)gid * 8 + 4 + i]);
        }
        float tot = 0.f;
        for (int r = 0; r < reps; ++r) {
            f32x4 acc[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
            for (int k = 0; k < 48; ++k) {  // 48 x 32 = K 1536
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
                asm volatile("" : "+v"(a[0]), "+v"(b[0]));  // keep the loop a loop
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) tot += acc[i][j][0] + acc[i][j][3];
        }
        if (tot == 123.456f) sink[gid] = tot;
    } else {
        bf16x8 a[2], b[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            a[i] = __builtin_bit_cast(bf16x8, src[(size_t)gid * 8 + i]);
            b[i] = __builtin_bit_cast(bf16x8, src[(size_t)gid * 8 + 4 + i]);
        }
        float tot = 0.f;
        for (int r = 0; r < reps; ++r) {
            f32x16 acc[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
#pragma unroll 4
            for (int k = 0; k < 96; ++k) {  // 96 x 16 = K 1536
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
                asm volatile("" : "+v"(a[0]), "+v"(b[0]));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) tot += acc[i][j][0] + acc[i][j][15];
        }
        if (tot == 123.456f) sink[gid] = tot;
    }
}

// Issue order of the 16 MFMAs of a k-step over a 4 x 4 block of accumulators: ORDER 0 keeps the A operand for four consecutive
// MFMAs (the GEMM tile's order), 1 keeps the B operand, 2 changes both on every instruction (diagonals)
template <int ORDER>
__global__ __launch_bounds__(256) void k_mfma_order(const uint4* __restrict__ src, float* __restrict__ sink, int reps) {
    const int gid = blockIdx.x * 256 + threadIdx.x;
    bf16x8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i] = __builtin_bit_cast(bf16x8, src[(size_t)gid * 8 + i]);
        b[i] = __builtin_bit_cast(bf16x8, src[(size_t)gid * 8 + 4 + i]);
    }
    float tot = 0.f;
    for (int r = 0; r < reps; ++r) {
        f32x4 acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
        for (int k = 0; k < 48; ++k) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int i = ORDER == 0 ? u : ORDER == 1 ? v : v, j = ORDER == 0 ? v : ORDER == 1 ? u : (u + v) & 3;
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            asm volatile("" : "+v"(a[0]), "+v"(b[0]));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) tot += acc[i][j][0] + acc[i][j][3];
    }
    if (tot == 123.456f) sink[gid] = tot;
}

// The same MFMA work fed from LDS the way a GEMM wave tile is: per k-step of 32 a wave reads TM A fragments and TN B
// fragments (ds_read_b128, conflict-free) and issues TM x TN MFMAs 16x16x32 -- wave tile (16 TM) x (16 TN).  NWAVES waves per
// workgroup, one workgroup per CU.  (8, 4) x 8 waves is the shipped 256x256 tile's wave layout.  The 4-wave cases (one wave
// per SIMD, compiler-placed LDS waits) are latency-bound at half the rate and say nothing about power: kept as the record
// of that.
template <int TM, int TN, int NWAVES>
__global__ __launch_bounds__(NWAVES * 64) void k_mfma_lds(const uint4* __restrict__ src, float* __restrict__ sink, int reps) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int BYTES = 64 * 1024;
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < BYTES / 16; i += NWAVES * 64) ((uint4*)lds)[i] = src[(size_t)blockIdx.x * (BYTES / 16) + i];
    __syncthreads();
    float tot = 0.f;
    const int wbase = (tid >> 6) * 4096;
    for (int r = 0; r < reps; ++r) {
        f32x4 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        bf16x8 a[2][TM], b[2][TN];
        auto rd = [&](int k, bf16x8* af, bf16x8* bfr) {
            const int base = (wbase + k * (TM + TN) * 1024) & (BYTES - 1);
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *(const bf16x8*)(lds + ((base + i * 1024 + lane * 16) & (BYTES - 1)));
#pragma unroll
            for (int j = 0; j < TN; ++j) bfr[j] = *(const bf16x8*)(lds + ((base + (TM + j) * 1024 + lane * 16) & (BYTES - 1)));
        };
        rd(0, a[0], b[0]);
#pragma unroll 2
        for (int k = 0; k < 48; ++k) {
            rd(k + 1, a[(k + 1) & 1], b[(k + 1) & 1]);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[k & 1][i], b[k & 1][j], acc[i][j], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) tot += acc[i][j][0] + acc[i][j][3];
    }
    if (tot == 123.456f) sink[blockIdx.x * NWAVES * 64 + tid] = tot;
}

static std::string slurp(const std::string& p) {
    FILE* f = fopen(p.c_str(), "r");
    if (!f) return "";
    char buf[4096];
    size_t n = fread(buf, 1, sizeof buf - 1, f);
    fclose(f);
    buf[n] = 0;
    return buf;
}

struct Sampler {
    std::string dev, hw;
    std::atomic<bool> on{false}, run{true};
    std::vector<double> mhz, watts;
    std::thread th;
    void start() {
        th = std::thread([this] {
            while (run) {
                if (on) {
                    std::string s = slurp(hw + "/freq1_input");
                    if (!s.empty()) mhz.push_back(atof(s.c_str()) / 1e6);
                    s = slurp(hw + "/power1_average");
                    if (s.empty()) s = slurp(hw + "/power1_input");
                    if (!s.empty()) watts.push_back(atof(s.c_str()) / 1e6);
                }
                usleep(20000);
            }
        });
    }
};

static double median(std::vector<double> v) {
    if (v.empty()) return 0;
    std::sort(v.begin(), v.end());
    return v[v.size() / 2];
}

static uint16_t bf16_of(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

int main() {
    CK(hipSetDevice(0));
    char bus[64] = {0};
    CK(hipDeviceGetPCIBusId(bus, sizeof bus, 0));
    for (char* p = bus; *p; ++p) *p = (char)tolower(*p);
    Sampler sm;
    if (DIR* d = opendir("/sys/class/drm")) {
        while (dirent* e = readdir(d)) {
            if (strncmp(e->d_name, "card", 4) != 0 || strchr(e->d_name, '-')) continue;
            std::string dev = std::string("/sys/class/drm/") + e->d_name + "/device";
            char real[1024];
            if (!realpath(dev.c_str(), real)) continue;
            std::string bus_noFn(bus);  // "0000:8b:00.0"
            if (strstr(real, bus_noFn.c_str())) sm.dev = dev;
        }
        closedir(d);
    }
    if (!sm.dev.empty()) {
        std::string hwroot = sm.dev + "/hwmon";
        if (DIR* d = opendir(hwroot.c_str())) {
            while (dirent* e = readdir(d))
                if (strncmp(e->d_name, "hwmon", 5) == 0) sm.hw = hwroot + "/" + e->d_name;
            closedir(d);
        }
    }
    printf("device %s -> sysfs %s (%s)\n", bus, sm.dev.c_str(), sm.hw.c_str());
    sm.start();

    hipDeviceProp_t pr;
    CK(hipGetDeviceProperties(&pr, 0));
    const int cus = pr.multiProcessorCount;
    const int blocks = cus * 2;  // 2 x 256 threads = 8 waves per CU, 2 per SIMD
    const size_t nthreads = (size_t)blocks * 256;
    std::vector<uint16_t> host(nthreads * 64);
    uint4* src;
    float* sink;
    CK(hipMalloc(&src, nthreads * 128));
    CK(hipMalloc(&sink, nthreads * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));

    const char* names[] = {"zeros", "+-1 (sign only)", "+-2^k (sign + exponent)", "normal(0,1) bf16", "uniform random bits (finite)"};
    for (int shape : {16, 32})
        for (int pat = 0; pat < 5; ++pat) {
            uint64_t st = 0x9E3779B97F4A7C15ull;
            auto rnd = [&] { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
            for (auto& h : host) {
                const uint64_t r = rnd();
                float g = 0.f;
                for (int t = 0; t < 12; ++t) g += (float)((rnd() >> 11) * (1.0 / 9007199254740992.0));
                g -= 6.f;  // ~ N(0, 1)
                switch (pat) {
                    case 0: h = 0; break;
                    case 1: h = (uint16_t)(0x3f80u | ((r & 1) << 15)); break;
                    case 2: h = (uint16_t)(((0x7cu + (r >> 1) % 8) << 7) | ((r & 1) << 15)); break;
                    case 3: h = bf16_of(g); break;
                    default: h = (uint16_t)((r & 0x807fu) | ((0x70u + (r >> 20) % 16) << 7)); break;
                }
            }
            CK(hipMemcpy(src, host.data(), nthreads * 128, hipMemcpyHostToDevice));
            const int reps = 200;
            auto launch = [&] {
                if (shape == 16) hipLaunchKernelGGL(k_mfma<16>, dim3(blocks), dim3(256), 0, 0, src, sink, reps);
                else hipLaunchKernelGGL(k_mfma<32>, dim3(blocks), dim3(256), 0, 0, src, sink, reps);
            };
            launch();
            CK(hipDeviceSynchronize());
            sm.mhz.clear(); sm.watts.clear();
            const auto t0 = std::chrono::steady_clock::now();
            int launches = 0;
            bool armed = false;
            CK(hipEventRecord(e0, 0));
            while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 2.5) {
                for (int i = 0; i < 10; ++i) launch();
                launches += 10;
                CK(hipDeviceSynchronize());
                if (!armed && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 0.5) { sm.on = true; armed = true; }
            }
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            sm.on = false;
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            // flops per launch: waves x reps x 16 accumulators-worth: 128 x 128(or 64 x 64) x 1536 x 2 per wave-rep
            const double per_wave_rep = shape == 16 ? 2.0 * 64 * 64 * 1536 : 2.0 * 64 * 64 * 1536;
            const double flops = (double)blocks * 4 * reps * per_wave_rep * launches;
            printf("mfma %-9s %-30s %8.1f TFLOP/s | sclk %6.0f MHz | %6.0f W | %d samples\n", shape == 16 ? "16x16x32" : "32x32x16", names[pat],
                   flops / (ms * 1e-3) / 1e12, median(sm.mhz), median(sm.watts), (int)sm.mhz.size());
            fflush(stdout);
        }
    // issue order (normal data, pattern 3 still in src? -- refill)
    {
        uint64_t st = 0x9E3779B97F4A7C15ull;
        auto rnd = [&] { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
        for (auto& h : host) {
            float g = 0.f;
            for (int t = 0; t < 12; ++t) g += (float)((rnd() >> 11) * (1.0 / 9007199254740992.0));
            h = bf16_of(g - 6.f);
        }
        CK(hipMemcpy(src, host.data(), nthreads * 128, hipMemcpyHostToDevice));
        const char* onames[] = {"A kept for 4 MFMAs", "B kept for 4 MFMAs", "both change every MFMA"};
        for (int rep2 = 0; rep2 < 2; ++rep2)
        for (int order = 0; order < 3; ++order) {
            const int reps = 200;
            auto launch = [&] {
                if (order == 0) hipLaunchKernelGGL(k_mfma_order<0>, dim3(blocks), dim3(256), 0, 0, src, sink, reps);
                else if (order == 1) hipLaunchKernelGGL(k_mfma_order<1>, dim3(blocks), dim3(256), 0, 0, src, sink, reps);
                else hipLaunchKernelGGL(k_mfma_order<2>, dim3(blocks), dim3(256), 0, 0, src, sink, reps);
            };
            launch();
            CK(hipDeviceSynchronize());
            sm.mhz.clear(); sm.watts.clear();
            const auto t0 = std::chrono::steady_clock::now();
            int launches = 0;
            bool armed = false;
            CK(hipEventRecord(e0, 0));
            while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 2.5) {
                for (int i = 0; i < 10; ++i) launch();
                launches += 10;
                CK(hipDeviceSynchronize());
                if (!armed && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 0.5) { sm.on = true; armed = true; }
            }
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            sm.on = false;
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double flops = (double)blocks * 4 * reps * (2.0 * 64 * 64 * 1536) * launches;
            printf("mfma 16x16x32 normal data, order: %-24s %8.1f TFLOP/s | sclk %6.0f MHz | %6.0f W\n", onames[order], flops / (ms * 1e-3) / 1e12,
                   median(sm.mhz), median(sm.watts));
            fflush(stdout);
        }
    }
    // LDS-fed wave tiles
    {
        uint4* lsrc;
        const size_t lbytes = (size_t)cus * 64 * 1024;
        CK(hipMalloc(&lsrc, lbytes));
        std::vector<uint16_t> lh(lbytes / 2);
        for (int pat : {0, 3}) {
            uint64_t st = 0x9E3779B97F4A7C15ull;
            auto rnd = [&] { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
            for (auto& h : lh) {
                float g = 0.f;
                for (int t = 0; t < 12; ++t) g += (float)((rnd() >> 11) * (1.0 / 9007199254740992.0));
                h = pat == 0 ? 0 : bf16_of(g - 6.f);
            }
            CK(hipMemcpy(lsrc, lh.data(), lbytes, hipMemcpyHostToDevice));
            for (int cfg = 0; cfg < 3; ++cfg) {
                const int reps = 100;
                const int tm = 8, tn = cfg == 1 ? 8 : 4, nw = cfg == 0 ? 8 : 4;
                auto launch = [&] {
                    if (cfg == 0) hipLaunchKernelGGL((k_mfma_lds<8, 4, 8>), dim3(cus), dim3(512), 64 * 1024, 0, lsrc, sink, reps);
                    else if (cfg == 1) hipLaunchKernelGGL((k_mfma_lds<8, 8, 4>), dim3(cus), dim3(256), 64 * 1024, 0, lsrc, sink, reps);
                    else hipLaunchKernelGGL((k_mfma_lds<8, 4, 4>), dim3(cus), dim3(256), 64 * 1024, 0, lsrc, sink, reps);
                };
                launch();
                CK(hipDeviceSynchronize());
                sm.mhz.clear(); sm.watts.clear();
                const auto t0 = std::chrono::steady_clock::now();
                int launches = 0;
                bool armed = false;
                CK(hipEventRecord(e0, 0));
                while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 2.5) {
                    for (int i = 0; i < 10; ++i) launch();
                    launches += 10;
                    CK(hipDeviceSynchronize());
                    if (!armed && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 0.5) { sm.on = true; armed = true; }
                }
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                sm.on = false;
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                const double flops = (double)cus * nw * reps * (2.0 * 16 * tm * 16 * tn * 1536) * launches;
                printf("lds-fed wave tile %3dx%-3d x %d waves/CU  %-18s %8.1f TFLOP/s | sclk %6.0f MHz | %6.0f W\n", 16 * tm, 16 * tn, nw, names[pat],
                       flops / (ms * 1e-3) / 1e12, median(sm.mhz), median(sm.watts));
                fflush(stdout);
            }
        }
    }
    sm.run = false;
    sm.th.join();
    return 0;
}
