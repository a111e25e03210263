// pq_encode.hip — GPU product-quantization encoder for rolled texture descriptors.
// Reference: TrainedPQEncoder.encode_multi, extraction/descriptor_PQ.py:19-27 (a loop over the 16 sub-spaces around
// scipy.cluster.vq.vq): codes[i][m] = index of the codeword of sub-quantizer m nearest to des[i][6m..6m+5].
// Arithmetic: squared L2 in fp32, d ascending, difference / product / sum rounded separately — the same function as the matcher's
// own ADC table entry (matching/include.h:327-359, adc.hip::lut_entry), first minimum on ties; so a point's code is exactly the
// codeword its own table ranks nearest.  The CPU restatement (oracle/afis_oracle.cpp::orc_pq_encode) agrees with scipy's vq on
// every golden case (tests/golden/golden_pq.npz).
//
// Work decomposition: a persistent workgroup of 16 waves keeps the whole codebook (16 x 256 x 6 fp32 = 96 KB) in LDS and walks
// tiles of 64 points; wave m owns sub-quantizer m, lane = point.  Every lane of a wave reads the same codeword (LDS broadcast,
// conflict-free); the descriptor tile is staged through LDS with coalesced 4-byte loads and a padded row (97 floats) so the
// per-lane reads of 6 consecutive floats hit distinct banks.  17 VALU + 1 compare + 2 selects per (point, codeword).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "afis_device.h"

namespace afis {

constexpr int kEncPts = 64;                   // points per tile
constexpr int kEncStride = kDes + 1;          // padded descriptor row

__global__ __launch_bounds__(1024) void k_pq_encode(const float* __restrict__ des, long long n, const float* __restrict__ codewords,
                                                    uint8_t* __restrict__ codes)
{
    __shared__ float s_cw[kM * kK * kDsub];                    // 96 KB
    __shared__ float s_des[kEncPts * kEncStride];              // 24.8 KB
    __shared__ uint32_t s_codes[kEncPts * kM / 4];             // 1 KB
    const int tid = threadIdx.x, lane = tid & 63;
    const int m = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < kM * kK * kDsub; i += 1024) s_cw[i] = codewords[i];
    const long long n_tiles = (n + kEncPts - 1) / kEncPts;
    for (long long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const long long p0 = tile * kEncPts;
        const int np = (int)((n - p0) < kEncPts ? (n - p0) : kEncPts);
        __syncthreads();                                       // codebook staged / previous tile's codes written out
        for (int i = tid; i < np * kDes; i += 1024) s_des[(i / kDes) * kEncStride + (i % kDes)] = des[p0 * kDes + i];
        __syncthreads();
        if (lane < np) {
            float x[kDsub];
#pragma unroll
            for (int d = 0; d < kDsub; ++d) x[d] = s_des[lane * kEncStride + m * kDsub + d];
            const float* cw = s_cw + m * kK * kDsub;
            float best = INFINITY; int arg = 0;
#pragma unroll 4
            for (int k = 0; k < kK; ++k) {
                float dist = 0.0f;
#pragma unroll
                for (int d = 0; d < kDsub; ++d) {
                    const float t = x[d] - cw[k * kDsub + d];
                    const float t2 = t * t;
                    dist += t2;
                }
                if (dist < best) { best = dist; arg = k; }     // strict: the first minimum wins
            }
            reinterpret_cast<uint8_t*>(s_codes)[lane * kM + m] = (uint8_t)arg;
        }
        __syncthreads();
        if (tid < np * kM / 4) reinterpret_cast<uint32_t*>(codes + p0 * kM)[tid] = s_codes[tid];
    }
}

hipError_t launch_pq_encode(const float* des, long long n, const float* codewords, uint8_t* codes, hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    const long long n_tiles = (n + kEncPts - 1) / kEncPts;
    const int grid = (int)(n_tiles < 256 ? n_tiles : 256);     // one persistent workgroup per CU (121 KB of LDS each)
    hipLaunchKernelGGL(k_pq_encode, dim3(grid), dim3(1024), 0, stream, des, n, codewords, codes);
    return hipGetLastError();
}

// Descriptors re-laid on the device as operand fragments of v_mfma_f32_16x16x4_f32 (minu.hip; layout: afis_ctx.h::fragment_tiles, which still does the latents' on the host):
// template t (rows off[t] .. off[t+1]) becomes ceil(n/16) tiles of 6 x 64 float4; lane l of load v holds des[16*tile + (l&15)][4*(4v + c) + (l>>4)], c = 0..3; rows past the
// template's end are zero.  One workgroup per template; the gallery's fragments (34 KB per template) no longer cross PCIe at commit.
__global__ __launch_bounds__(256) void k_fragment_tiles(const float* __restrict__ des, const int32_t* __restrict__ off, const int32_t* __restrict__ tile_off, float4* __restrict__ frag)
{
    const int t = blockIdx.x;
    const int r0 = off[t], n = off[t + 1] - r0;
    const int t0 = tile_off[t], nt = tile_off[t + 1] - t0;
    for (int e = threadIdx.x; e < nt * 6 * 64; e += 256) {                // one float4 of the output per thread and step
        const int tile = e / (6 * 64), r = e - tile * (6 * 64), v = r >> 6, l = r & 63;
        const int row = tile * 16 + (l & 15), lg = l >> 4;
        float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < n) {
            const float* src = des + (size_t)(r0 + row) * kDes + lg;
            x = make_float4(src[4 * (4 * v + 0)], src[4 * (4 * v + 1)], src[4 * (4 * v + 2)], src[4 * (4 * v + 3)]);
        }
        frag[(size_t)(t0 + tile) * (6 * 64) + r] = x;
    }
}

hipError_t launch_fragment_tiles(const float* des, const int32_t* off, const int32_t* tile_off, int n_templates, void* frag, hipStream_t stream)
{
    if (n_templates <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_fragment_tiles, dim3(n_templates), dim3(256), 0, stream, des, off, tile_off, (float4*)frag);
    return hipGetLastError();
}

}  // namespace afis
