// xchg.hip -- the one exchange step of the head-sharded decode path: every rank's attention output
// [rows, row_bytes] lands in every rank's memory, by direct stores into the peers' exchange areas (xGMI peer memory mapped with
// hipIpc; one process per GPU) followed by a flag.  One launch per layer per token, no host involvement, nothing the hipGraph of
// the token step cannot hold.  The reference has no distributed code; this replaces the all_gather_into_tensor the eager sharded
// step issues in front of o_proj (cuda_supported_gear/modeling_llamagear.py:478-482 is where the heads are merged).
//
// Exchange area of a rank (uncached device memory, so neither the writer's nor the owner's L2 ever holds a line of it):
//   word 0              epoch counter of THIS rank (advanced by the kernel: graph replays need no host argument)
//   words 64 + 64 p + r flag "rank r's slice of parity p is complete", value = the epoch it belongs to
//   byte 1024 ...       data [2 parities][world][bytes_per_rank]
// Two parities: a rank can be at most one exchange ahead of its slowest peer (it cannot finish exchange e + 1 before every peer
// has pushed e + 1, which a peer does only after it has left exchange e), so slot (e & 1) is never overwritten while it is read.
#include <string.h>

#include "common.h"

#define XCHG_HEADER 1024
#define XCHG_MAX_WORLD 64
#define XCHG_SPIN_TICKS 300000000ull   // 3 s of the 100 MHz wall clock: a lost peer turns into a status word, not a hang

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct XchgArgs {
    const u32x4* src;
    char* const* peers;   // device array [world]: base of every rank's exchange area as mapped in THIS process
    u32x4* out;
    uint32_t* status;
    int world, rank, rows;
    uint32_t row_chunks;   // 16-byte chunks per row
};

__device__ __forceinline__ void st_sys16(u32x4* p, u32x4 v) {
    // write-through system-scope store (the area is uncached; sc0 sc1 also keeps the store out of the local L2 for peer memory)
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ u32x4 ld_sys16(const u32x4* p) {
    u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

__global__ __launch_bounds__(256) void xchg_allgather_kernel(XchgArgs a) {
    __shared__ uint32_t s_epoch;
    __shared__ uint32_t s_bad;
    const int tid = threadIdx.x;
    char* own = a.peers[a.rank];
    if (tid == 0) {
        uint32_t e = __hip_atomic_load((uint32_t*)own, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) + 1u;
        if (e == 0u) e = 2u;   // (never 0: flags start at 0; parity of 2 == parity of the wrapped 0)
        __hip_atomic_store((uint32_t*)own, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        s_epoch = e;
        // a rank that gave up once does not wait again (every later exchange would cost the full time limit)
        s_bad = a.status ? __hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    }
    __syncthreads();
    const uint32_t e = s_epoch;
    const uint32_t par = e & 1u;
    const uint32_t nchunk = (uint32_t)a.rows * a.row_chunks;            // chunks of one rank's slice
    const size_t slice = (size_t)nchunk * 16u;
    const size_t slot = XCHG_HEADER + ((size_t)par * a.world + a.rank) * slice;

    // push: my slice into slot (parity, rank) of every rank's area (my own included)
    for (uint32_t c = tid; c < nchunk; c += 256u) {
        const u32x4 v = a.src[c];
        for (int p = 0; p < a.world; ++p) st_sys16((u32x4*)(a.peers[p] + slot) + c, v);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");       // system scope: every store above has left this GPU
    __syncthreads();
    if (tid < a.world) {
        uint32_t* f = (uint32_t*)a.peers[tid] + 64 + 64 * par + a.rank;
        __hip_atomic_store(f, e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // wait for every rank's slice of this epoch
    if (tid < a.world && !s_bad) {
        const uint32_t* f = (const uint32_t*)own + 64 + 64 * par + tid;
        const uint64_t t0 = wall_clock64();
        while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != e) {
            __builtin_amdgcn_s_sleep(8);
            if (wall_clock64() - t0 > XCHG_SPIN_TICKS) {
                s_bad = 1u;
                break;
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    __syncthreads();
    if (s_bad && tid == 0 && a.status) atomicOr(a.status, 1u);
    // copy out: area [parity][r][row][chunk] -> out [row][r][chunk] (rank r's heads at slot r of every row)
    const u32x4* data = (const u32x4*)(own + XCHG_HEADER + (size_t)par * a.world * slice);
    const uint32_t total = nchunk * (uint32_t)a.world;
    for (uint32_t i = tid; i < total; i += 256u) {
        const uint32_t r = i / nchunk, w = i - r * nchunk;
        const uint32_t row = w / a.row_chunks, c = w - row * a.row_chunks;
        a.out[((size_t)row * a.world + r) * a.row_chunks + c] = ld_sys16(data + i);
    }
}

extern "C" size_t gear_xchg_bytes(int world, size_t bytes_per_rank) {
    if (world < 1 || world > XCHG_MAX_WORLD) return 0;
    return XCHG_HEADER + 2u * (size_t)world * bytes_per_rank;
}

extern "C" int gear_xchg_alloc(size_t bytes, void** ptr) {
    GEAR_CHECK_ARG(ptr && bytes >= XCHG_HEADER, "gear_xchg_alloc: bytes must come from gear_xchg_bytes()");
    void* p = nullptr;
    hipError_t e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached);
    if (e != hipSuccess) {
        gear_set_error("gear_xchg_alloc: hipExtMallocWithFlags(%zu, uncached): %s", bytes, hipGetErrorString(e));
        return -2;
    }
    e = hipMemset(p, 0, bytes);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        (void)hipFree(p);
        gear_set_error("gear_xchg_alloc: clearing the area: %s", hipGetErrorString(e));
        return -2;
    }
    *ptr = p;
    return 0;
}

extern "C" int gear_xchg_free(void* ptr) {
    if (!ptr) return 0;
    hipError_t e = hipFree(ptr);
    if (e != hipSuccess) {
        gear_set_error("gear_xchg_free: %s", hipGetErrorString(e));
        return -2;
    }
    return 0;
}

extern "C" int gear_xchg_export(const void* ptr, void* handle64) {
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "the boundary carries the handle as 64 opaque bytes");
    GEAR_CHECK_ARG(ptr && handle64, "gear_xchg_export: null argument");
    hipIpcMemHandle_t h;
    hipError_t e = hipIpcGetMemHandle(&h, const_cast<void*>(ptr));
    if (e != hipSuccess) {
        gear_set_error("gear_xchg_export: hipIpcGetMemHandle: %s", hipGetErrorString(e));
        return -2;
    }
    memcpy(handle64, &h, 64);
    return 0;
}

extern "C" int gear_xchg_open(const void* handle64, void** ptr) {
    GEAR_CHECK_ARG(ptr && handle64, "gear_xchg_open: null argument");
    hipIpcMemHandle_t h;
    memcpy(&h, handle64, 64);
    void* p = nullptr;
    hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) {
        gear_set_error("gear_xchg_open: hipIpcOpenMemHandle: %s", hipGetErrorString(e));
        return -2;
    }
    *ptr = p;
    return 0;
}

extern "C" int gear_xchg_close(void* ptr) {
    if (!ptr) return 0;
    hipError_t e = hipIpcCloseMemHandle(ptr);
    if (e != hipSuccess) {
        gear_set_error("gear_xchg_close: %s", hipGetErrorString(e));
        return -2;
    }
    return 0;
}

extern "C" int gear_xchg_allgather(const void* src, int rows, size_t row_bytes, int world, int rank, const void* peers, void* out,
                                   void* status, void* stream) {
    GEAR_CHECK_ARG(src && peers && out, "gear_xchg_allgather: null argument");
    GEAR_CHECK_ARG(world >= 1 && world <= XCHG_MAX_WORLD && rank >= 0 && rank < world, "gear_xchg_allgather: rank %d of %d", rank, world);
    GEAR_CHECK_ARG(rows >= 1 && row_bytes >= 16 && row_bytes % 16 == 0, "gear_xchg_allgather: rows %d x %zu bytes (rows of whole 16-byte chunks)",
                   rows, row_bytes);
    GEAR_CHECK_ARG((uint64_t)rows * row_bytes * (uint64_t)world <= (64u << 20), "gear_xchg_allgather: %d x %zu bytes x %d ranks is not a decode-step exchange",
                   rows, row_bytes, world);
    XchgArgs a;
    a.src = (const u32x4*)src;
    a.peers = (char* const*)peers;
    a.out = (u32x4*)out;
    a.status = (uint32_t*)status;
    a.world = world;
    a.rank = rank;
    a.rows = rows;
    a.row_chunks = (uint32_t)(row_bytes / 16);
    hipLaunchKernelGGL(xchg_allgather_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, a);
    GEAR_CHECK_LAUNCH("gear_xchg_allgather");
    return 0;
}
