// seeded_kernels.hip.h -- seeded ("compressed") server keys expanded on the device: the bodies travel over PCIe
// (P22: 24 MB instead of 104 MB of keys, P44: 1 GB instead of 3.7 GB), the masks -- the compression seed's
// AES-128 counter-mode stream, see csrc/seeded_keys.cpp for the reference files -- are generated where they are
// needed.  One thread per AES block: block a = AES_seed(a as a little-endian u128); the stream starts at byte 1 of
// block 0, so mask word w is bytes [1 + 8w, 9 + 8w): words 2a and 2a+1 come out of block a, the last byte of word
// 2a+1 is byte 0 of block a+1 (the neighbouring lane's, or recomputed at a wavefront's edge).
// Mask word g = row * mask_per_row + c lands at row * row_words + c of the standard-domain key; bodies fill the rest
// of each row (seeded_scatter_bodies_kernel).  S-box in LDS, byte-wise rounds: 10^8 blocks for the largest key,
// far from being a bottleneck.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fhe {

struct SeededExpandArgs {
    uint8_t round_keys[11][16];
    uint64_t* out;            // standard-domain key
    uint64_t rows;            // ciphertexts in the key
    uint32_t mask_per_row;    // n (keyswitch key) or k N (bootstrap key)
    uint32_t row_words;       // n + 1 or (k + 1) N
};

__device__ __forceinline__ uint8_t seeded_xtime(uint8_t x) { return (uint8_t)((x << 1) ^ ((x >> 7) * 0x1b)); }

__device__ __forceinline__ void seeded_aes_block(const uint8_t (*rk)[16], const uint8_t* sbox, uint64_t counter, uint8_t s[16]) {
#pragma unroll
    for (int i = 0; i < 16; i++) s[i] = (uint8_t)((i < 8 ? (counter >> (8 * i)) : 0) ^ rk[0][i]);   // key sizes here stay below 2^64 blocks
#pragma unroll 1
    for (int r = 1; r <= 10; r++) {
        uint8_t t[16];
#pragma unroll
        for (int c = 0; c < 4; c++)
#pragma unroll
            for (int row = 0; row < 4; row++) t[4 * c + row] = sbox[s[4 * ((c + row) & 3) + row]];
        if (r < 10) {
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const uint8_t a0 = t[4 * c], a1 = t[4 * c + 1], a2 = t[4 * c + 2], a3 = t[4 * c + 3];
                s[4 * c] = (uint8_t)(seeded_xtime(a0) ^ (seeded_xtime(a1) ^ a1) ^ a2 ^ a3);
                s[4 * c + 1] = (uint8_t)(a0 ^ seeded_xtime(a1) ^ (seeded_xtime(a2) ^ a2) ^ a3);
                s[4 * c + 2] = (uint8_t)(a0 ^ a1 ^ seeded_xtime(a2) ^ (seeded_xtime(a3) ^ a3));
                s[4 * c + 3] = (uint8_t)((seeded_xtime(a0) ^ a0) ^ a1 ^ a2 ^ seeded_xtime(a3));
            }
        } else {
#pragma unroll
            for (int i = 0; i < 16; i++) s[i] = t[i];
        }
#pragma unroll
        for (int i = 0; i < 16; i++) s[i] ^= rk[r][i];
    }
}

__global__ void __launch_bounds__(256) seeded_expand_kernel(SeededExpandArgs a, const uint8_t* __restrict__ sbox_in) {
    __shared__ uint8_t sbox[256];
    __shared__ uint8_t rk[11][16];
    sbox[threadIdx.x] = sbox_in[threadIdx.x];
    if (threadIdx.x < 176) rk[threadIdx.x / 16][threadIdx.x % 16] = a.round_keys[threadIdx.x / 16][threadIdx.x % 16];
    __syncthreads();
    const uint64_t total_words = a.rows * a.mask_per_row;
    const uint64_t blocks = (1 + 8 * total_words + 15) / 16;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    // `base` is workgroup-uniform: every lane runs every iteration (lanes past the end compute and store nothing),
    // so the shuffle below always has its neighbour
    for (uint64_t base = (uint64_t)blockIdx.x * blockDim.x; base < blocks; base += stride) {
        const uint64_t blk = base + threadIdx.x;
        uint8_t s[16];
        seeded_aes_block(rk, sbox, blk, s);
        // byte 0 of the next block: the next lane's, except at the wavefront's last lane
        unsigned next0 = (unsigned)__shfl_down((int)s[0], 1);
        if ((threadIdx.x & 63) == 63) {
            uint8_t nb[16];
            seeded_aes_block(rk, sbox, blk + 1, nb);
            next0 = nb[0];
        }
        uint64_t w0 = 0, w1 = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) w0 |= (uint64_t)s[1 + i] << (8 * i);
#pragma unroll
        for (int i = 0; i < 7; i++) w1 |= (uint64_t)s[9 + i] << (8 * i);
        w1 |= (uint64_t)(next0 & 0xFF) << 56;
        const uint64_t g0 = 2 * blk, g1 = g0 + 1;
        if (g0 < total_words) a.out[(g0 / a.mask_per_row) * a.row_words + g0 % a.mask_per_row] = w0;
        if (g1 < total_words) a.out[(g1 / a.mask_per_row) * a.row_words + g1 % a.mask_per_row] = w1;
    }
}

// bodies [rows][body_per_row] -> the tail of every row of the standard-domain key
__global__ void __launch_bounds__(256) seeded_scatter_bodies_kernel(const uint64_t* __restrict__ bodies, uint64_t* __restrict__ out,
                                                                    uint64_t rows, uint32_t mask_per_row, uint32_t row_words) {
    const uint32_t body_per_row = row_words - mask_per_row;
    const uint64_t total = rows * body_per_row;
    for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (uint64_t)gridDim.x * blockDim.x)
        out[(e / body_per_row) * row_words + mask_per_row + e % body_per_row] = bodies[e];
}

// Seeded LWE ciphertexts (shortint CompressedCiphertext, one compression seed EACH -- shortint/ciphertext/mod.rs:471-478,
// seeded_lwe_ciphertext_decompression.rs:11-50): one 64-thread workgroup per ciphertext expands that seed's key
// schedule in LDS, then its lwe_dim mask words; the body goes last.  An encrypted 256-char string arrives as 94 KB
// of (seed, body) pairs instead of 16.8 MB of ciphertexts.
__global__ void __launch_bounds__(64) seeded_lwe_expand_kernel(const uint8_t* __restrict__ seeds /* [count][16] */,
                                                               const uint64_t* __restrict__ bodies, const uint8_t* __restrict__ sbox_in,
                                                               uint64_t* __restrict__ out, uint32_t lwe_dim) {
    __shared__ uint8_t sbox[256];
    __shared__ uint8_t rk[11][16];
    for (int i = threadIdx.x; i < 256; i += 64) sbox[i] = sbox_in[i];
    __syncthreads();
    const uint32_t ct = blockIdx.x;
    if (threadIdx.x == 0) {           // AES-128 key schedule of this ciphertext's seed
        for (int i = 0; i < 16; i++) rk[0][i] = seeds[(size_t)ct * 16 + i];
        uint8_t rcon = 1;
        for (int r = 1; r <= 10; r++) {
            uint8_t t[4] = {sbox[rk[r - 1][13]], sbox[rk[r - 1][14]], sbox[rk[r - 1][15]], sbox[rk[r - 1][12]]};
            t[0] ^= rcon;
            rcon = seeded_xtime(rcon);
            for (int c = 0; c < 4; c++)
                for (int b = 0; b < 4; b++) rk[r][4 * c + b] = rk[r - 1][4 * c + b] ^ (c == 0 ? t[b] : rk[r][4 * (c - 1) + b]);
        }
    }
    __syncthreads();
    uint64_t* dst = out + (size_t)ct * (lwe_dim + 1);
    const uint32_t blocks = (1 + 8 * lwe_dim + 15) / 16;
    for (uint32_t base = 0; base < blocks; base += 64) {
        const uint32_t blk = base + threadIdx.x;
        uint8_t s[16];
        seeded_aes_block(rk, sbox, blk, s);
        unsigned next0 = (unsigned)__shfl_down((int)s[0], 1);
        if (threadIdx.x == 63) {
            uint8_t nb[16];
            seeded_aes_block(rk, sbox, (uint64_t)blk + 1, nb);
            next0 = nb[0];
        }
        uint64_t w0 = 0, w1 = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) w0 |= (uint64_t)s[1 + i] << (8 * i);
#pragma unroll
        for (int i = 0; i < 7; i++) w1 |= (uint64_t)s[9 + i] << (8 * i);
        w1 |= (uint64_t)(next0 & 0xFF) << 56;
        if (2 * blk < lwe_dim) dst[2 * blk] = w0;
        if (2 * blk + 1 < lwe_dim) dst[2 * blk + 1] = w1;
    }
    if (threadIdx.x == 0) dst[lwe_dim] = bodies[ct];
}

}  // namespace fhe
