// libtvae_hip.so, the streaming kernels of the hot path (rotated bank, attention head, coordinate
// transform, skinny decoder / encoder ends, likelihoods, Adam) behind their C-ABI entry points (include/tvae_hip.h).
#include "abi_common.hpp"
#include "small_kernels.hpp"
#include "rotate_bank_kernels.hpp"
#include "fused_tail_kernels.hpp"

using namespace tvae;

template <int NO>
static void launch_heads_fwd(const float* W, const float* X, long ldx, const float* bias, float* Y, long ldy, int C,
                             long N, int vec, hipStream_t st) {
    hipLaunchKernelGGL(heads_fwd_kernel<NO>, dim3(panels_of(N, 1024)), dim3(256), 0, st, W, X, ldx, bias, Y, ldy, C, N,
                       vec);
}
template <int NO>
static void launch_heads_bwd(const float* W, const float* dY, long ldy, const float* X, long ldx, float* dX, long lddx,
                             int C, long N, int act, float slope, float* part, int vec, hipStream_t st) {
    hipLaunchKernelGGL(heads_bwd_kernel<NO>, dim3(panels_of(N, PANEL8)), dim3(256), 0, st, W, dY, ldy, X, ldx, dX, lddx,
                       C, N, act, slope, part, vec);
}

extern "C" {

int tvae_abi_version(void) { return 7; }

int tvae_rotate_bank_fwd(const float* weight, const int* tap_idx, const float* tap_w, float* bank, int C, int Cin,
                         int ksz, int R, tvae_stream_t stream) {
    const int k2 = ksz * ksz;
    const long total = (long)C * R * Cin * k2;
    if (total >= 2147483647L || !aligned16(tap_idx) || !aligned16(tap_w)) return (int)hipErrorInvalidValue;
    const long npair = (long)C * Cin;
    if ((npair + RB_CH - 1) / RB_CH > 65535) return (int)hipErrorInvalidValue;
    const int npx32 = (ksz + 31) / 32, npy8 = (ksz + 7) / 8;
    hipLaunchKernelGGL(rotate_bank_fwd_kernel, dim3((unsigned)(R * npx32 * npy8), (unsigned)((npair + RB_CH - 1) / RB_CH)),
                       dim3(256), 0, S(stream), weight, tap_idx, tap_w, bank, C, Cin, ksz, R);
    TVAE_CHECK_LAUNCH();
    return 0;
}

int tvae_rotate_bank_bwd(const float* dbank, const int* csr_ptr, const int* csr_r, const int* csr_dst,
                         const float* csr_w, float* dweight, int C, int Cin, int ksz, int R, int accumulate,
                         tvae_stream_t stream) {
    const int k2 = ksz * ksz;
    const long total = (long)C * Cin * k2;
    if (total * R >= 2147483647L) return (int)hipErrorInvalidValue;
    const long npair = (long)C * Cin;
    if (k2 > (1 << 24) || R > 127) return (int)hipErrorInvalidValue;
    // pairs per thread: four where that still leaves >= 2 workgroups per CU, else two
    const int npx = (ksz + 7) / 8;
    const bool big = (long)npx * npx * ((npair + 15) / 16) >= 512;                 // (2 x the 256 CUs of an MI355X)
    const int ppw = big ? 16 : 8;
    if ((npair + ppw - 1) / ppw > 65535) return (int)hipErrorInvalidValue;
    const dim3 grid((unsigned)(npx * npx), (unsigned)((npair + ppw - 1) / ppw));
    if (big)
        hipLaunchKernelGGL(rotate_bank_bwd_kernel<4>, grid, dim3(256), 0, S(stream), dbank, csr_ptr, csr_r, csr_dst, csr_w, dweight,
                           C, Cin, ksz, R, accumulate);
    else
        hipLaunchKernelGGL(rotate_bank_bwd_kernel<2>, grid, dim3(256), 0, S(stream), dbank, csr_ptr, csr_r, csr_dst, csr_w, dweight,
                           C, Cin, ksz, R, accumulate);
    TVAE_CHECK_LAUNCH();
    return 0;
}
int tvae_rowdot_seg(const float* X, long ldx, const float* V, int no, int M, int N, int seglen, float* out, float* amax,
                    tvae_stream_t stream) {
    if (seglen <= 0 || M <= 0) return (int)hipErrorInvalidValue;
    const int nseg = (N + seglen - 1) / seglen;
    dim3 grid(M, nseg), block(256);
    if (!V) no = 1;
    switch (no) {
        case 1: hipLaunchKernelGGL(rowdot_seg_kernel<1>, grid, block, 0, S(stream), X, ldx, V, N, seglen, out, M, amax); break;
        case 2: hipLaunchKernelGGL(rowdot_seg_kernel<2>, grid, block, 0, S(stream), X, ldx, V, N, seglen, out, M, amax); break;
        case 3: hipLaunchKernelGGL(rowdot_seg_kernel<3>, grid, block, 0, S(stream), X, ldx, V, N, seglen, out, M, amax); break;
        case 4: hipLaunchKernelGGL(rowdot_seg_kernel<4>, grid, block, 0, S(stream), X, ldx, V, N, seglen, out, M, amax); break;
        default: return (int)hipErrorInvalidValue;
    }
    TVAE_CHECK_LAUNCH();
    return 0;
}

int tvae_seg_sum(const float* in, int S_, long L, float* out, float scale, int accumulate, tvae_stream_t stream) {
    if (L <= 64 && S_ >= 32)         // few outputs, many segments: one wave per output
        hipLaunchKernelGGL(seg_sum_wave_kernel, dim3((unsigned)L), dim3(64), 0, S(stream), in, S_, L, out, scale, accumulate);
    else if (S_ >= 64 && L <= 65536)      // many outputs, many segments: 64 outputs x 16 segment lanes per workgroup
        hipLaunchKernelGGL(seg_sum_tile_kernel, dim3((unsigned)((L + 63) / 64)), dim3(1024), 0, S(stream), in, S_, L, out, scale,
                           accumulate);
    else
        hipLaunchKernelGGL(seg_sum_kernel, dim3(grid1d(L, 256)), dim3(256), 0, S(stream), in, S_, L, out, scale, accumulate);
    TVAE_CHECK_LAUNCH();
    return 0;
}

int tvae_elbo_reduce(const float* lp, const float* kl, int B, double* elbo, float* logp, double* kld, tvae_stream_t stream) {
    if (B <= 0 || !lp || !kl || !elbo || !logp || !kld) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(elbo_reduce_kernel, dim3(1), dim3(256), 0, S(stream), lp, kl, B, elbo, logp, kld);
    TVAE_CHECK_LAUNCH();
    return 0;
}
int tvae_elbo_reduce_bwd(const double* g_elbo, const float* g_logp, const double* g_kld, int B, float* g_lp, float* g_kl,
                         tvae_stream_t stream) {
    if (B <= 0 || !g_lp || !g_kl) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(elbo_reduce_bwd_kernel, dim3((B + 255) / 256), dim3(256), 0, S(stream), g_elbo, g_logp, g_kld, B, g_lp, g_kl);
    TVAE_CHECK_LAUNCH();
    return 0;
}

int tvae_coldot(const float* X, long ldx, int M, int N, const float* W, int wsm, int wso, const float* bias, int no,
                float* out, tvae_stream_t stream) {
    dim3 grid((N + 255) / 256), block(256);
    const size_t sh = (size_t)M * no * sizeof(float);
    switch (no) {
        case 1: hipLaunchKernelGGL(coldot_kernel<1>, grid, block, sh, S(stream), X, ldx, M, N, W, wsm, wso, bias, out); break;
        case 2: hipLaunchKernelGGL(coldot_kernel<2>, grid, block, sh, S(stream), X, ldx, M, N, W, wsm, wso, bias, out); break;
        case 3: hipLaunchKernelGGL(coldot_kernel<3>, grid, block, sh, S(stream), X, ldx, M, N, W, wsm, wso, bias, out); break;
        case 4: hipLaunchKernelGGL(coldot_kernel<4>, grid, block, sh, S(stream), X, ldx, M, N, W, wsm, wso, bias, out); break;
        default: return (int)hipErrorInvalidValue;
    }
    TVAE_CHECK_LAUNCH();
    return 0;
}

int tvae_outer_mask(const float* dy, int no, const float* W, int wsm, int wso, const float* H, long ldh, float* D,
                    long ldd, int M, int N, int act, float slope, tvae_stream_t stream) {
    dim3 grid((N + 255) / 256, (M + 15) / 16), block(256);
    switch (no) {
        case 1: hipLaunchKernelGGL(outer_mask_kernel<1>, grid, block, 0, S(stream), dy, W, wsm, wso, H, ldh, D, ldd, M, N, act, slope); break;
        case 2: hipLaunchKernelGGL(outer_mask_kernel<2>, grid, block, 0, S(stream), dy, W, wsm, wso, H, ldh, D, ldd, M, N, act, slope); break;
        case 3: hipLaunchKernelGGL(outer_mask_kernel<3>, grid, block, 0, S(stream), dy, W, wsm, wso, H, ldh, D, ldd, M, N, act, slope); break;
        case 4: hipLaunchKernelGGL(outer_mask_kernel<4>, grid, block, 0, S(stream), dy, W, wsm, wso, H, ldh, D, ldd, M, N, act, slope); break;
        default: return (int)hipErrorInvalidValue;
    }
    TVAE_CHECK_LAUNCH();
    return 0;
}

// ---- fused tails of the two MLPs (fused_tail_kernels.hpp) --------------------------------------------------

int tvae_dec_out_bwd(const float* gy, int n_out, const float* Wo, const float* H, long ldh, float* D, long ldd, int F,
                     long N, int act, float slope, float* part, long part_floats, float* tot, tvae_stream_t stream) {
    if (F <= 0 || N <= 0) return 0;
    const int np = panels_of(N, PANEL16);
    if (n_out < 1 || n_out > 4 || part_floats < (long)np * F * (1 + n_out)) return (int)hipErrorInvalidValue;
    const int vec = (aligned16(H) && (!D || aligned16(D)) && ldh % 4 == 0 && ldd % 4 == 0) ? 1 : 0;   // D may be NULL
    // row slices: enough workgroups for ~4 per CU, at least 16 rows (four per wave) each
    int fs = np >= 1024 ? 1 : (1024 + np - 1) / np;
    if (fs > F / 16) fs = F / 16 > 0 ? F / 16 : 1;
    dim3 grid(np, fs), block(256);
    switch (n_out) {
        case 1: hipLaunchKernelGGL(dec_out_bwd_kernel<1>, grid, block, 0, S(stream), gy, Wo, H, ldh, D, ldd, F, N, act, slope, part, vec); break;
        case 2: hipLaunchKernelGGL(dec_out_bwd_kernel<2>, grid, block, 0, S(stream), gy, Wo, H, ldh, D, ldd, F, N, act, slope, part, vec); break;
        case 3: hipLaunchKernelGGL(dec_out_bwd_kernel<3>, grid, block, 0, S(stream), gy, Wo, H, ldh, D, ldd, F, N, act, slope, part, vec); break;
        default: hipLaunchKernelGGL(dec_out_bwd_kernel<4>, grid, block, 0, S(stream), gy, Wo, H, ldh, D, ldd, F, N, act, slope, part, vec); break;
    }
    TVAE_CHECK_LAUNCH();
    hipLaunchKernelGGL(part_total_kernel, dim3(F), dim3(256), 0, S(stream), (const float*)part, np, F, 1 + n_out, tot);
    TVAE_CHECK_LAUNCH();
    return 0;
}

int tvae_dec_in_bwd(const float* d, long ldd, const float* xr, const float* Wc, int F, int B, int Np, float* gxr,
                    float* Simg, float* dbc, float* dWc, float* part, long part_floats, tvae_stream_t stream) {
    if (F <= 0 || B <= 0 || Np <= 0) return 0;
    const int cpi = panels_of(Np, PANEL16);
    if (part_floats < (long)B * cpi * F * 3) return (int)hipErrorInvalidValue;
    const int vec = (aligned16(d) && ldd % 4 == 0 && Np % 4 == 0) ? 1 : 0;
    hipLaunchKernelGGL(dec_in_bwd_kernel, dim3(B * cpi), dim3(256), 0, S(stream), d, ldd, xr, Wc, F, Np, cpi, gxr, part,
                       vec);
    TVAE_CHECK_LAUNCH();
    return tvae_dec_in_total(part, B, cpi, F, Simg, dbc, dWc, stream);
}

int tvae_heads_fwd(const float* W, const float* X, long ldx, const float* bias, float* Y, long ldy, int nh, int C,
                   long N, tvae_stream_t stream) {
    if (N <= 0 || C <= 0) return 0;
    if (nh < 1 || nh > 8) return (int)hipErrorInvalidValue;
    const int vec = (aligned16(X) && aligned16(Y) && ldx % 4 == 0 && ldy % 4 == 0) ? 1 : 0;
    switch (nh) {
        case 1: launch_heads_fwd<1>(W, X, ldx, bias, Y, ldy, C, N, vec, S(stream)); break;
        case 2: launch_heads_fwd<2>(W, X, ldx, bias, Y, ldy, C, N, vec, S(stream)); break;
        case 3: launch_heads_fwd<3>(W, X, ldx, bias, Y, ldy, C, N, vec, S(stream)); break;
        case 4: launch_heads_fwd<4>(W, X, ldx, bias, Y, ldy, C, N, vec, S(stream)); break;
        case 5: launch_heads_fwd<5>(W, X, ldx, bias, Y, ldy, C, N, vec, S(stream)); break;
        case 6: launch_heads_fwd<6>(W, X, ldx, bias, Y, ldy, C, N, vec, S(stream)); break;
        case 7: launch_heads_fwd<7>(W, X, ldx, bias, Y, ldy, C, N, vec, S(stream)); break;
        default: launch_heads_fwd<8>(W, X, ldx, bias, Y, ldy, C, N, vec, S(stream)); break;
    }
    TVAE_CHECK_LAUNCH();
    return 0;
}

int tvae_heads_bwd(const float* W, const float* dY, long ldy, const float* X, long ldx, float* dX, long lddx, int nh,
                   int C, long N, int act, float slope, float* part, long part_floats, float* tot,
                   tvae_stream_t stream) {
    if (N <= 0 || C <= 0) return 0;
    const int np = panels_of(N, PANEL8);
    if (nh < 1 || nh > 8 || part_floats < (long)np * C * (nh + 1)) return (int)hipErrorInvalidValue;
    const int vec = (aligned16(X) && (!dX || aligned16(dX)) && aligned16(dY) && ldx % 4 == 0 && lddx % 4 == 0 && ldy % 4 == 0) ? 1 : 0;
    switch (nh) {
        case 1: launch_heads_bwd<1>(W, dY, ldy, X, ldx, dX, lddx, C, N, act, slope, part, vec, S(stream)); break;
        case 2: launch_heads_bwd<2>(W, dY, ldy, X, ldx, dX, lddx, C, N, act, slope, part, vec, S(stream)); break;
        case 3: launch_heads_bwd<3>(W, dY, ldy, X, ldx, dX, lddx, C, N, act, slope, part, vec, S(stream)); break;
        case 4: launch_heads_bwd<4>(W, dY, ldy, X, ldx, dX, lddx, C, N, act, slope, part, vec, S(stream)); break;
        case 5: launch_heads_bwd<5>(W, dY, ldy, X, ldx, dX, lddx, C, N, act, slope, part, vec, S(stream)); break;
        case 6: launch_heads_bwd<6>(W, dY, ldy, X, ldx, dX, lddx, C, N, act, slope, part, vec, S(stream)); break;
        case 7: launch_heads_bwd<7>(W, dY, ldy, X, ldx, dX, lddx, C, N, act, slope, part, vec, S(stream)); break;
        default: launch_heads_bwd<8>(W, dY, ldy, X, ldx, dX, lddx, C, N, act, slope, part, vec, S(stream)); break;
    }
    TVAE_CHECK_LAUNCH();
    // totals of the panels: two coalesced stages when the caller's workspace has room for the PT_GROUPS partials behind them
    const long used = (long)np * C * (nh + 1), mv = (long)C * (nh + 1);
    if (np >= 4 * PT_GROUPS && mv % 4 == 0 && part_floats >= ((used + 3) & ~3L) + PT_GROUPS * mv && aligned16(part)) {
        float* partial = part + ((used + 3) & ~3L);
        hipLaunchKernelGGL(part_total_s1_kernel, dim3(PT_GROUPS), dim3(256), 0, S(stream), (const float*)part, np, (int)(mv / 4),
                           partial);
        TVAE_CHECK_LAUNCH();
        hipLaunchKernelGGL(part_total_s2_kernel, dim3((unsigned)((mv + 255) / 256)), dim3(256), 0, S(stream),
                           (const float*)partial, C, nh + 1, tot);
        TVAE_CHECK_LAUNCH();
        return 0;
    }
    hipLaunchKernelGGL(part_total_kernel, dim3(C), dim3(256), 0, S(stream), (const float*)part, np, C, nh + 1, tot);
    TVAE_CHECK_LAUNCH();
    return 0;
}

int tvae_act_bwd(const float* dY, const float* Y, float* dpre, long n, int act, float slope, tvae_stream_t stream) {
    hipLaunchKernelGGL(act_bwd_kernel, dim3(grid1d(n, 256)), dim3(256), 0, S(stream), dY, Y, dpre, n, act, slope);
    TVAE_CHECK_LAUNCH();
    return 0;
}

static HeadParams make_head(const float* heads, long ldh, const float* E, const float* eps_z, const float* eps_t,
                            const float* p_r, const float* off, const float* p_tr, const float* grid, int R, int P,
                            int zd, float sigma_p, float theta_off_scale) {
    HeadParams hp;
    hp.heads = heads; hp.ldh = ldh; hp.E = E; hp.eps_z = eps_z; hp.eps_t = eps_t;
    hp.p_r = p_r; hp.off = off; hp.p_tr = p_tr; hp.grid = grid;
    hp.R = R; hp.P = P; hp.zd = zd; hp.sigma_p = sigma_p; hp.theta_off_scale = theta_off_scale;
    return hp;
}

// Few images with very many positions (cfg5: 8 x 266 256): G workgroups per image, partial results through `part`.
// With many images (B >= 128) one workgroup per image already fills the chip.
static int head_chunks(int B, int RP, int zd, long part_floats, int per_chunk_floats, int& chunk) {
    chunk = RP;
    if (B >= 128 || RP < 16384 || zd > 64) return 1;
    int G = (RP + 4095) / 4096;
    const int cap = 1024 / (B > 0 ? B : 1);
    if (G > cap) G = cap;
    if (G < 2 || part_floats < (long)B * G * per_chunk_floats) return 1;
    chunk = (RP + G - 1) / G;
    return (RP + chunk - 1) / chunk;
}

int tvae_attn_head_fwd(const float* heads, long ldh, const float* E, const float* eps_z, const float* eps_t,
                       const float* p_r, const float* off, const float* p_tr, const float* grid, int B, int R, int P,
                       int zd, float sigma_p, float theta_off_scale, float* attn, float* q, float* a, float* z,
                       float* theta, float* dx, float* kl, float* part, long part_floats, tvae_stream_t stream) {
    if (B <= 0) return 0;
    HeadParams hp = make_head(heads, ldh, E, eps_z, eps_t, p_r, off, p_tr, grid, R, P, zd, sigma_p, theta_off_scale);
    int chunk;
    const int G = part ? head_chunks(B, R * P, zd, part_floats, head_part_floats(zd), chunk) : 1;
    if (G > 1) {
        hipLaunchKernelGGL(attn_head_fwd_a_kernel, dim3(B * G), dim3(1024), 0, S(stream), hp, G, chunk, attn, part);
        TVAE_CHECK_LAUNCH();
        hipLaunchKernelGGL(attn_head_fwd_b_kernel, dim3(B * G), dim3(1024), 0, S(stream), hp, G, chunk, (const float*)attn,
                           q, a, part);
        TVAE_CHECK_LAUNCH();
        hipLaunchKernelGGL(attn_head_fwd_c_kernel, dim3(B), dim3(256), 0, S(stream), hp, B, G, (const float*)part, z, theta,
                           dx, kl);
        TVAE_CHECK_LAUNCH();
        return 0;
    }
    hipLaunchKernelGGL(attn_head_fwd_kernel, dim3(B), dim3(1024), 0, S(stream), hp, attn, q, a, z, theta, dx, kl);
    TVAE_CHECK_LAUNCH();
    return 0;
}

int tvae_attn_head_bwd(const float* heads, long ldh, const float* q, const float* a, const float* eps_z,
                       const float* eps_t, const float* p_r, const float* off, const float* p_tr, const float* grid,
                       int B, int R, int P, int zd, float sigma_p, float theta_off_scale, const float* gz,
                       const float* gth, const float* gdx, const float* gkl, const float* g_attn, const float* g_q,
                       const float* g_a, float* dheads, float* part, long part_floats, tvae_stream_t stream) {
    if (B <= 0) return 0;
    HeadParams hp = make_head(heads, ldh, nullptr, eps_z, eps_t, p_r, off, p_tr, grid, R, P, zd, sigma_p,
                              theta_off_scale);
    int chunk;
    const int G = part ? head_chunks(B, R * P, zd, part_floats, 2, chunk) : 1;
    if (G > 1) {
        hipLaunchKernelGGL(attn_head_bwd_a_kernel, dim3(B * G), dim3(1024), 0, S(stream), hp, G, chunk, q, a, gz, gth, gdx,
                           gkl, g_q, g_a, part);
        TVAE_CHECK_LAUNCH();
        hipLaunchKernelGGL(attn_head_bwd_b_kernel, dim3(B * G), dim3(1024), 0, S(stream), hp, G, chunk, q, a, gz, gth, gdx,
                           gkl, g_attn, g_q, g_a, (const float*)part, dheads);
        TVAE_CHECK_LAUNCH();
        return 0;
    }
    hipLaunchKernelGGL(attn_head_bwd_kernel, dim3(B), dim3(1024), 0, S(stream), hp, q, a, gz, gth, gdx, gkl, g_attn,
                       g_q, g_a, dheads);
    TVAE_CHECK_LAUNCH();
    return 0;
}

int tvae_rot_pool_fwd(const float* A1, const float* fw, const float* fb, float* X, int C, int B, int R, int P,
                      tvae_stream_t stream) {
    if ((long)C * B * P <= 0) return 0;
    if (R < 1 || R > 16) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(rot_pool_fwd_kernel, dim3(grid1d((long)C * B * P, 256)), dim3(256), 0, S(stream), A1, fw, fb, X, C, B,
                       R, P);
    TVAE_CHECK_LAUNCH();
    return 0;
}

int tvae_rot_pool_bwd(const float* A1, const float* dX, const float* fw, float* dA1, float* part, long part_floats,
                      float* dtot, int C, int B, int R, int P, int act, float slope, tvae_stream_t stream) {
    if ((long)C * B * P <= 0) return 0;
    const int nb = grid1d((long)C * B * P, 256, 1024);
    if (R < 1 || R > 16 || part_floats < (long)(R + 1) * nb) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(rot_pool_bwd_kernel, dim3(nb), dim3(256), 0, S(stream), A1, dX, fw, dA1, part, C, B, R, P, act, slope);
    TVAE_CHECK_LAUNCH();
    // dtot[0..R) = d fw, dtot[R] = d fb: the per-block partials [nb][R + 1] summed in block order
    hipLaunchKernelGGL(seg_sum_kernel, dim3(1), dim3(64), 0, S(stream), (const float*)part, nb, (long)(R + 1), dtot, 1.f, 0);
    TVAE_CHECK_LAUNCH();
    return 0;
}

int tvae_get_latent(const float* heads, long ldh, const float* p_r, const float* off, const float* grid, int B, int R,
                    int P, int zd, float theta_off_scale, float* zc, float* theta_mu, float* dx, tvae_stream_t stream) {
    if (B <= 0) return 0;
    HeadParams hp = make_head(heads, ldh, nullptr, nullptr, nullptr, p_r, off, nullptr, grid, R, P, zd, 1.f,
                              theta_off_scale);
    hipLaunchKernelGGL(get_latent_kernel, dim3(B), dim3(1024), 0, S(stream), hp, zc, theta_mu, dx);
    TVAE_CHECK_LAUNCH();
    return 0;
}

int tvae_coord_fwd(const float* xc, const float* dx, const float* theta, float* xr, int B, int Np,
                   tvae_stream_t stream) {
    hipLaunchKernelGGL(coord_fwd_kernel, dim3(grid1d((long)B * Np, 256)), dim3(256), 0, S(stream), xc, dx, theta, xr,
                       B, Np);
    TVAE_CHECK_LAUNCH();
    return 0;
}

int tvae_coord_bwd(const float* xc, const float* dx, const float* theta, const float* gxr, float* gdx, float* gtheta,
                   int B, int Np, tvae_stream_t stream) {
    if (B <= 0) return 0;
    hipLaunchKernelGGL(coord_bwd_kernel, dim3(B), dim3(256), 0, S(stream), xc, dx, theta, gxr, gdx, gtheta, Np);
    TVAE_CHECK_LAUNCH();
    return 0;
}

int tvae_dec_l0_fwd(const float* xr, const float* Wc, const float* bc, const float* LB, float* h, long ldh, int F,
                    long Ntot, int Np, int act, float slope, tvae_stream_t stream) {
    dim3 grid((unsigned)((Ntot + 255) / 256), (F + 15) / 16), block(256);
    hipLaunchKernelGGL(dec_l0_fwd_kernel, grid, block, 0, S(stream), xr, Wc, bc, LB, h, ldh, F, Ntot, Np, act, slope);
    TVAE_CHECK_LAUNCH();
    return 0;
}

int tvae_latent_bias(const float* Wl, const float* z, float* LB, int B, int F, int zd, tvae_stream_t stream) {
    hipLaunchKernelGGL(latent_bias_kernel, dim3((B * F + 255) / 256), dim3(256), 0, S(stream), Wl, z, LB, B, F, zd);
    TVAE_CHECK_LAUNCH();
    return 0;
}

int tvae_latent_bwd(const float* S_, const float* Wl, const float* z, float* dWl, float* dz, int B, int F, int zd,
                    tvae_stream_t stream) {
    const int tot = (F * zd > B * zd) ? F * zd : B * zd;
    if (tot <= 0) return 0;
    hipLaunchKernelGGL(latent_bwd_kernel, dim3((tot + 63) / 64), dim3(1024), 0, S(stream), S_, Wl, z, dWl, dz, B, F, zd);
    TVAE_CHECK_LAUNCH();
    return 0;
}

int tvae_fourier_fwd(const float* xr, const float* Wf, const float* bf, float sigma, float* feat, long ld, int F,
                     long Ntot, tvae_stream_t stream) {
    dim3 grid((unsigned)((Ntot + 255) / 256), (F + 15) / 16), block(256);
    hipLaunchKernelGGL(fourier_fwd_kernel, grid, block, 0, S(stream), xr, Wf, bf, sigma, feat, ld, F, Ntot);
    TVAE_CHECK_LAUNCH();
    return 0;
}

int tvae_fourier_bwd(const float* xr, const float* Wf, const float* bf, float sigma, const float* dfeat, long ld,
                     int F, long Ntot, float* gxr, tvae_stream_t stream) {
    hipLaunchKernelGGL(fourier_bwd_kernel, dim3((unsigned)((Ntot + 255) / 256)), dim3(256), 0, S(stream), xr, Wf, bf,
                       sigma, dfeat, ld, F, Ntot, gxr);
    TVAE_CHECK_LAUNCH();
    return 0;
}

int tvae_loglik_fwd(const float* yh, const float* y, float* lp, int B, int L, int kind, tvae_stream_t stream) {
    if (B <= 0) return 0;
    // few images with many pixels (galaxy: 8 x 49 152): sixteen waves per image instead of four (125 -> ~35 us); the per-image
    // sum is a fixed tree either way
    const int bt = (B < 256 && L >= 16384) ? 1024 : 256;
    hipLaunchKernelGGL(loglik_fwd_kernel, dim3(B), dim3(bt), 0, S(stream), yh, y, lp, L, kind);
    TVAE_CHECK_LAUNCH();
    return 0;
}

int tvae_loglik_bwd(const float* yh, const float* y, const float* glp, float* gyh, int B, int L, int kind,
                    tvae_stream_t stream) {
    hipLaunchKernelGGL(loglik_bwd_kernel, dim3(grid1d((long)B * L, 256)), dim3(256), 0, S(stream), yh, y, glp, gyh, B,
                       L, kind);
    TVAE_CHECK_LAUNCH();
    return 0;
}

int tvae_ctf_corr(const float* in, const float* ctf, float* out, int B, int n, int kc, int flip, tvae_stream_t stream) {
    if (B <= 0) return 0;
    if ((kc & 1) == 0) return (int)hipErrorInvalidValue;
    dim3 grid((n * n + 255) / 256, B), block(256);
    hipLaunchKernelGGL(ctf_corr_kernel, grid, block, 0, S(stream), in, ctf, out, n, kc, flip);
    TVAE_CHECK_LAUNCH();
    return 0;
}

int tvae_loglik_masked_fwd(const float* yh, const float* y, const float* dx, float inv_spacing, float radius, int B,
                           int n, float* lp, tvae_stream_t stream) {
    if (B <= 0) return 0;
    hipLaunchKernelGGL(loglik_masked_fwd_kernel, dim3(B), dim3(256), 0, S(stream), yh, y, dx, inv_spacing, radius, n,
                       lp);
    TVAE_CHECK_LAUNCH();
    return 0;
}

int tvae_loglik_masked_bwd(const float* yh, const float* y, const float* dx, float inv_spacing, float radius, int B,
                           int n, const float* glp, float* gyh, tvae_stream_t stream) {
    hipLaunchKernelGGL(loglik_masked_bwd_kernel, dim3(grid1d((long)B * n * n, 256)), dim3(256), 0, S(stream), yh, y, dx,
                       inv_spacing, radius, n, glp, gyh, B);
    TVAE_CHECK_LAUNCH();
    return 0;
}

int tvae_adam_flat(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps,
                   float bc1, float bc2_sqrt, float grad_scale, tvae_stream_t stream) {
    hipLaunchKernelGGL(adam_flat_kernel, dim3(grid1d(n, 256, 2048)), dim3(256), 0, S(stream), p, g, m, v, n, lr, b1,
                       b2, eps, bc1, bc2_sqrt, grad_scale);
    TVAE_CHECK_LAUNCH();
    return 0;
}
int tvae_dec_in_total(float* part, int B, int cpi, int F, float* Simg, float* dbc, float* dWc, tvae_stream_t stream) {
    // second stage of the fused first-layer backward: part[B*cpi panels][F][3] -> per-image sums, bias and weight grads.
    // part is CONSUMED: the first panel of every image is overwritten with the image's sums.
    if (B <= 0 || F <= 0 || cpi <= 0) return 0;
    hipLaunchKernelGGL(dec_in_total_img_kernel, dim3(B), dim3(256), 0, S(stream), part, cpi, F, Simg);
    TVAE_CHECK_LAUNCH();
    hipLaunchKernelGGL(dec_in_total_sum_kernel, dim3((3 * F + 63) / 64), dim3(256), 0, S(stream), (const float*)part, B, cpi, F,
                       dbc, dWc);
    TVAE_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
