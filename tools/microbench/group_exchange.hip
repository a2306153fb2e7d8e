// Debug aid (not product): what does a per-step exchange between the K workgroups of a group cost on gfx950?
//   hipcc --offload-arch=gfx950 -O3 -o tools/microbench/group_exchange tools/microbench/group_exchange.hip
//   tools/microbench/group_exchange
// The question behind it (batch512): a recurrent step of a 32-row tile is one CU's affair (~10 us: 5 us of MFMA
// issue by 8 waves on 4 SIMDs + gate phase + barriers).  Splitting the tile's hidden units over K workgroups divides the
// MFMA issue by K but makes every step end with an exchange of the new h slices through memory: each workgroup writes its
// slice (32 rows x 256 / K units x 4 B of h2), releases, arrives at the group's counter, waits for the other K - 1, acquires,
// and reads their slices.  This measures exactly that loop, with nothing else in it, for K = 2, 4, 8 and for groups whose
// members share an XCD (blockIdx = member * 8 * groups_per_xcd ... i.e. same blockIdx % 8) or are spread over the XCDs
// (consecutive blockIdx).  Spins are bounded: a group that does not meet within ~50 ms gives up and reports it.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// MODE 1, the same exchange WITHOUT the two agent-scope fences: every exchanged word is written and read with a relaxed
// agent-scope atomic access (global_store / global_load with sc1: coherent per access, no L2 write-back / invalidate of
// everything else the workgroup has touched), ordered by "stores acknowledged (s_waitcnt vmcnt(0)) -> barrier -> arrive" on
// the writing side and "counter seen -> barrier -> loads issued" on the reading side.
// one step: write own slice, fence, arrive, wait, fence, read the other slices (summed into `sink` so nothing is optimised away)
template <int K, int MODE>
__global__ __launch_bounds__(128) void exchange_kernel(uint4* __restrict__ h, unsigned* __restrict__ counters, int steps, int spread,
                                                       int n_groups, unsigned long long* __restrict__ cycles, int* __restrict__ gave_up,
                                                       float* __restrict__ sink) {
    // spread = 1: members of a group have consecutive blockIdx (different XCDs); 0: same blockIdx % 8 (one XCD)
    const int bid = blockIdx.x;
    int group, member;
    if (spread) { group = bid / K; member = bid % K; }
    else {
        // blocks with equal bid % 8 share an XCD: lay the groups out so that a group's K members are bid = base + 8 * j
        const int xcd = bid & 7, q = bid >> 3;           // q-th block of this XCD
        group = (q / K) * 8 + xcd;
        member = q % K;
    }
    if (group >= n_groups) return;
    constexpr int SLICE16 = 32 * (256 / K) * 4 / 16;      // 16-byte chunks of one member's slice
    const int tid = threadIdx.x;
    uint4* mine = h + ((size_t)group * K + member) * SLICE16 * 2;       // two parities
    unsigned* cnt = counters + group * 32;                                 // own 128-byte line
    unsigned acc = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    bool ok = true;
    for (int s = 0; s < steps && ok; ++s) {
        uint4* dst = mine + (s & 1) * SLICE16;
        if (MODE == 0) {
            for (int i = tid; i < SLICE16; i += 128) dst[i] = make_uint4(s, member, i, acc);
            __threadfence();                                                // release (agent scope)
        } else if (MODE == 2) {
            // as MODE 1 with 16-byte buffer accesses carrying the sc1 bit (what lstm_rec_h2_split_kernel issues)
            typedef unsigned u4 __attribute__((ext_vector_type(4)));
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(dst, 0, 0x7fffffff, 0x00020000);
            for (int i = tid; i < SLICE16; i += 128) {
                const u4 v = {(unsigned)s, (unsigned)member, (unsigned)i, acc};
                __builtin_amdgcn_raw_buffer_store_b128(v, rs, i * 16u, 0, 16);
            }
            __builtin_amdgcn_s_waitcnt(0);
        } else {
            unsigned long long* d64 = reinterpret_cast<unsigned long long*>(dst);
            for (int i = tid; i < 2 * SLICE16; i += 128)
                __hip_atomic_store(d64 + i, (unsigned long long)(unsigned)s | ((unsigned long long)(unsigned)(i + acc) << 32),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_s_waitcnt(0);                                  // every store acknowledged
        }
        __syncthreads();
        if (tid == 0) {
            atomicAdd(cnt, 1u);
            const unsigned want = (unsigned)K * (unsigned)(s + 1);
            int spins = 0;
            while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                if (++spins > (1 << 22)) { ok = false; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            if (!ok) atomicAdd(gave_up, 1);
        }
        ok = __syncthreads_and(ok);
        if (MODE == 0) __threadfence();                                     // acquire
        for (int m = 0; m < K; ++m) {
            if (m == member) continue;
            const uint4* src = h + ((size_t)group * K + m) * SLICE16 * 2 + (s & 1) * SLICE16;
            // (all loads of a slice in flight together, checked afterwards: a check per load would put a full memory round
            // trip between one load and the next)
            bool stale = false;
            if (MODE == 0) {
                constexpr int PER = SLICE16 / 128;
                uint4 v[PER];
#pragma unroll
                for (int j = 0; j < PER; ++j) v[j] = src[tid + 128 * j];
#pragma unroll
                for (int j = 0; j < PER; ++j) {
                    acc += v[j].x + v[j].z;
                    stale |= v[j].x != (unsigned)s;
                }
            } else if (MODE == 2) {
                constexpr int PER = SLICE16 / 128;
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(src), 0, 0x7fffffff, 0x00020000);
                typedef unsigned u4 __attribute__((ext_vector_type(4)));
                u4 v[PER];
#pragma unroll
                for (int j = 0; j < PER; ++j) v[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, (tid + 128 * j) * 16u, 0, 16);
#pragma unroll
                for (int j = 0; j < PER; ++j) {
                    acc += v[j].x + v[j].z;
                    stale |= v[j].x != (unsigned)s;
                }
            } else {
                constexpr int PER = 2 * SLICE16 / 128;
                const unsigned long long* s64 = reinterpret_cast<const unsigned long long*>(src);
                unsigned long long v[PER];
#pragma unroll
                for (int j = 0; j < PER; ++j) v[j] = __hip_atomic_load(s64 + tid + 128 * j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (int j = 0; j < PER; ++j) {
                    acc += (unsigned)(v[j] >> 32);
                    stale |= (unsigned)v[j] != (unsigned)s;
                }
            }
            if (stale) atomicAdd(gave_up, 1 << 16);                         // a stale slice: the exchange is broken
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) cycles[bid] = t1 - t0;
    if (acc == 0xdeadbeef) sink[0] = 1.0f;
}

template <int K, int MODE>
int run(int groups, int steps) {
    uint4* h = nullptr;
    unsigned* counters = nullptr;
    unsigned long long* cycles = nullptr;
    int* gave_up = nullptr;
    float* sink = nullptr;
    const int blocks = (groups * K + 7) / 8 * 8;
    CHECK(hipMalloc(&h, (size_t)groups * K * 2 * 32 * (256 / K) * 4));
    CHECK(hipMalloc(&counters, (size_t)groups * 128));
    CHECK(hipMalloc(&cycles, blocks * 8));
    CHECK(hipMalloc(&gave_up, 4));
    CHECK(hipMalloc(&sink, 4));
    for (int spread = 0; spread < 2; ++spread) {
        CHECK(hipMemset(counters, 0, (size_t)groups * 128));
        CHECK(hipMemset(gave_up, 0, 4));
        CHECK(hipMemset(cycles, 0, blocks * 8));
        hipEvent_t a, b;
        CHECK(hipEventCreate(&a));
        CHECK(hipEventCreate(&b));
        CHECK(hipEventRecord(a, 0));
        hipLaunchKernelGGL((exchange_kernel<K, MODE>), dim3(blocks), dim3(128), 0, 0, h, counters, steps, spread, groups, cycles, gave_up, sink);
        CHECK(hipEventRecord(b, 0));
        CHECK(hipEventSynchronize(b));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, a, b));
        int bad = 0;
        CHECK(hipMemcpy(&bad, gave_up, 4, hipMemcpyDeviceToHost));
        printf("%s K=%d groups=%d (%d workgroups) members %s: %.2f us per step%s\n",
               MODE == 2 ? "sc1 16-byte buffer ops: " : (MODE ? "sc1 accesses, no fences:" : "agent-scope fences:     "), K, groups, groups * K,
               spread ? "spread over the XCDs" : "on one XCD       ", 1e3 * ms / steps,
               bad ? (bad >> 16 ? "  ** STALE DATA SEEN **" : "  ** a group gave up **") : "");
    }
    hipFree(h); hipFree(counters); hipFree(cycles); hipFree(gave_up); hipFree(sink);
    return 0;
}

int main(int argc, char** argv) {
    const int steps = argc > 1 ? atoi(argv[1]) : 2000;
    // 512 windows = 16 tiles x 2 directions = 32 groups; 1024 windows = 64 groups
    for (int groups : {32, 64}) {
        if (run<2, 0>(groups, steps)) return 1;
        if (run<4, 0>(groups, steps)) return 1;
        if (groups * 8 <= 256 && run<8, 0>(groups, steps)) return 1;
        if (run<2, 1>(groups, steps)) return 1;
        if (run<4, 1>(groups, steps)) return 1;
        if (groups * 8 <= 256 && run<8, 1>(groups, steps)) return 1;
        if (run<2, 2>(groups, steps)) return 1;
        if (run<4, 2>(groups, steps)) return 1;
        if (groups * 8 <= 256 && run<8, 2>(groups, steps)) return 1;
    }
    return 0;
}
