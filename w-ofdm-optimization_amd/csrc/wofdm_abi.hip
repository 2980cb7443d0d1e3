// wofdm_abi.hip -- extern "C" boundary of libwofdm_hip.so (declared in include/wofdm.h).
// Plans own the constants in HBM; launches are asynchronous on the caller's stream.
#include "../../include/wofdm.h"
#include "wofdm_kernel.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <utility>
#include <vector>

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(e_ == hipErrorOutOfMemory ? WOFDM_E_NOMEM : WOFDM_E_HIP, "%s: %s",    \
                        #expr, hipGetErrorString(e_));                                        \
    } while (0)

struct geom {
    int N, k, S, mu, rho, beta, delta, gamma, kappa, L, P, B, T, NL;
};

int check_cfg(const wofdm_cfg *c, geom *g)
{
    if (!c) return fail(WOFDM_E_INVALID, "cfg is NULL");
    const int N = c->n_fft;
    if (N != 64 && N != 128 && N != 256 && N != 512 && N != 1024)
        return fail(WOFDM_E_UNSUPPORTED, "n_fft=%d not in {64,128,256,512,1024}", N);
    if (c->bits_per_sc != 2 && c->bits_per_sc != 4 && c->bits_per_sc != 6)
        return fail(WOFDM_E_UNSUPPORTED, "bits_per_sc=%d not in {2,4,6}", c->bits_per_sc);
    if (c->syms_per_frame < 2 || c->syms_per_frame > WOFDM_MAX_SYMS)
        return fail(WOFDM_E_UNSUPPORTED, "syms_per_frame=%d not in [2,%d]", c->syms_per_frame,
                    WOFDM_MAX_SYMS);
    if (c->n_taps < 1 || c->n_taps > WOFDM_MAX_TAPS)
        return fail(WOFDM_E_UNSUPPORTED, "n_taps=%d not in [1,%d]", c->n_taps, WOFDM_MAX_TAPS);
    if (c->cp < 0 || c->cs < 0 || c->tail_tx < 0 || c->tail_rx < 0 || (c->tail_rx & 1) ||
        c->prefix_rm < 0 || c->circ_shift < 0 || c->circ_shift >= N)
        return fail(WOFDM_E_INVALID, "negative length, odd tail_rx or circ_shift >= n_fft");
    if (c->cp > N || c->cs > N || c->tail_rx > N)
        return fail(WOFDM_E_INVALID, "cp, cs and tail_rx must not exceed n_fft");
    if (c->n_channels < 1 || c->n_snr < 1 || c->n_window_pairs < 1)
        return fail(WOFDM_E_INVALID, "n_channels, n_snr, n_window_pairs must be >= 1");
    g->N = N; g->k = c->bits_per_sc; g->S = c->syms_per_frame;
    g->mu = c->cp; g->rho = c->cs; g->beta = c->tail_tx; g->delta = c->tail_rx;
    g->gamma = c->prefix_rm; g->kappa = c->circ_shift; g->L = c->n_taps;
    g->P = N + g->mu + g->rho; g->B = g->P - g->beta; g->T = g->beta + g->S * g->B;
    g->NL = c->noise_before_truncate ? g->T + g->L - 1 : g->S * g->B;
    if (2 * g->beta > g->P) return fail(WOFDM_E_INVALID, "2*tail_tx exceeds the symbol length");
    // the Rx reshape of matlab/main_BER_calculation.m:262-263 needs B == N + delta + gamma
    if (g->B != N + g->delta + g->gamma)
        return fail(WOFDM_E_INVALID, "n_fft+cp+cs-tail_tx (%d) != n_fft+tail_rx+prefix_rm (%d)",
                    g->B, N + g->delta + g->gamma);
    if (g->mu + g->rho > wofdm_cpcs_max(N) || g->beta > 16 || g->delta > 64)
        return fail(WOFDM_E_UNSUPPORTED, "kernel limits: cp+cs <= %d, tail_tx <= 16, tail_rx <= 64",
                    wofdm_cpcs_max(N));
    if (g->B > 64 * wofdm_rb(N))
        return fail(WOFDM_E_UNSUPPORTED, "cp+cs-tail_tx=%d exceeds the 64 samples the kernel's "
                    "FIR tiling allows", g->B - N);
    const uint64_t cells = (uint64_t)c->n_channels * c->n_snr * c->n_window_pairs;
    if (cells >= (1u << 28)) return fail(WOFDM_E_UNSUPPORTED, "more than 2^28 cells");
    return WOFDM_OK;
}

}  // namespace

struct wofdm_plan {
    int device = 0;
    wofdm_cfg cfg{};
    geom g{};
    uint32_t n_cells = 0;
    float *d_wtx = nullptr, *d_wrx = nullptr, *d_nlin = nullptr;
    float2 *d_h = nullptr;
    int *d_geo = nullptr;
    float2 *d_nscr = nullptr;          // unit-noise scratch rows, one per workgroup (large DFTs)
    uint64_t nscr_elems = 0;           // float2 elements allocated there
    wofdm_kparams base{};
    wofdm_kernel_fn fn[4] = {nullptr, nullptr, nullptr, nullptr};   // kernels of the variant in use
    int var = WOFDM_VAR_PLAIN;         // kernel variant in use (configure())
    bool has_alloc = false, has_mask = false;
    // wofdm_plan_set_option: diagnostic kernel choice
    bool force_direct_mask = false;    // WOFDM_OPT_TXMASK_DIRECT: never the FFT form
    bool fir_valu = false;             // WOFDM_OPT_FIR_VALU: FIR on the VALU in every layout
    bool dft_valu = false;             // WOFDM_OPT_DFT_VALU: N = 256 transforms on the VALU (layouts 6, 7 instead of 10, 11)
    int max_spw = 0;                   // WOFDM_OPT_MAX_SPW: 0 = no cap
    uint32_t *d_amask = nullptr;       // [N/4] words, byte r bit 7: subcarrier j + r N/4 not loaded
    float2 *d_tmask = nullptr;         // [2P-1] circular impulse response of the Tx mask
    float2 *d_tspec = nullptr;         // [WOFDM_TXFFT_LEN] its fast-convolution spectrum (FFT form)
    float *d_tspec4 = nullptr;         // the same x 2^4 in layout 15's order: [kc][re | im][lane][j] <-> bin lane + 64 j + 256 kc
    unsigned *d_status = nullptr;      // kernel status word (wofdm_kparams::status)
    uint4 *d_fira = nullptr;           // [n_ch][4][64] Toeplitz operands of the matrix-pipe FIR (MFMA A layout)
    float firm_sx = 1.f, firm_sh = 1.f; // powers of two carried by the f16 samples / f16 taps there
    uint4 *d_dftc = nullptr;           // [10][64] operands of the matrix-pipe 256-point transforms (build_dftc)
    float *d_rxs = nullptr;            // [n_snr][n_ch] power of two that centres the received samples in the f16 range
#ifdef WOFDM_AUDIT
    uint32_t *audit = nullptr;
    uint32_t audit_items = 0;
#endif
    int occ = 1, cus = 1, spw = 1;     // spw: layout id of the kernels in use (wofdm_spw)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

namespace {

// words of the allocation table (wofdm_plan_set_allocation: N/4 in the plain order, then the layouts' own orders)
size_t wofdm_amask_words(int N)
{
    const size_t NQ = (size_t)N / 4;
    return N == 256 ? 2 * NQ + 256 : NQ + (NQ > 256 ? NQ : 256);
}

// Select the kernel variant from the plan's options and size everything that depends on it
// (symbols per wave, LDS bytes, frame buffer length, occupancy).  Plan state changes only on success.
int configure(wofdm_plan *pl)
{
    const geom &g = pl->g;
    // the mask runs as fast convolution where its transform length allows, else in direct form
    const bool fft_ok = g.N <= WOFDM_TXFFT_MAX_N && 3 * g.P - 2 <= WOFDM_TXFFT_LEN && !pl->force_direct_mask;
    const int var = pl->has_mask ? (fft_ok ? WOFDM_VAR_TXFFT : WOFDM_VAR_TXMASK)
                                 : (pl->has_alloc ? WOFDM_VAR_ALLOC : WOFDM_VAR_PLAIN);
    const bool masked = var == WOFDM_VAR_TXMASK || var == WOFDM_VAR_TXFFT;
    const bool firm = !pl->fir_valu;
    const bool mdft = !pl->dft_valu;
    int spw = masked ? wofdm_spw_masked(g.N, g.B, firm) : wofdm_spw(g.N, g.S, g.B, true, firm, mdft);
    // (layout 15: the fast-convolution mask at N = 256 with all four transforms on the matrix pipe)
    if (var == WOFDM_VAR_TXFFT && g.N == 256 && spw == 9 && mdft && WOFDM_TXFFT_LEN == 1024) spw = 15;
    if (pl->max_spw > 0 && wofdm_nsym(spw, g.N) > pl->max_spw)
        spw = (pl->max_spw == 1) ? 1 : wofdm_spw(g.N, g.S, g.B, false);
    // (WOFDM_DEV_LDS_PAD: developer builds only -- unused LDS bytes per workgroup, to measure how the rate depends on the
    // number of workgroups a CU holds; never defined in the shipped library)
#ifndef WOFDM_DEV_LDS_PAD
#define WOFDM_DEV_LDS_PAD 0
#endif
    const unsigned lds = wofdm_lds_bytes(g.N, g.T, spw, g.S, g.B)
                         + (var == WOFDM_VAR_TXMASK ? wofdm_txmask_lds_bytes(g.N) : 0u)
                         + (var == WOFDM_VAR_TXFFT && spw != 15 ? wofdm_txfft_lds_bytes() : 0u) + (unsigned)(WOFDM_DEV_LDS_PAD);
    if (lds > 160u * 1024u)
        return fail(WOFDM_E_UNSUPPORTED, "frame needs %u bytes of LDS (160 KiB per workgroup)", lds);
    wofdm_kernel_fn fn[4];
    for (int m = 0; m < 4; ++m) {
        fn[m] = wofdm_select_kernel(g.N, g.k, spw, m, var);
        if (!fn[m])
            return fail(WOFDM_E_UNSUPPORTED, "no kernel for n_fft=%d bits_per_sc=%d variant %d%s", g.N, g.k,
                        var, masked ? " (the Tx mask needs n_fft <= 512)" : "");
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(fn[m]),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    int occ = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(
        &occ, reinterpret_cast<const void *>(fn[WOFDM_MODE_GEN]), 64 * wofdm_waves(spw, g.N, g.S, g.B), lds));
    if (occ < 1) return fail(WOFDM_E_UNSUPPORTED, "kernel does not fit a CU (LDS %u bytes)", lds);
    // (the LDS goes in units of 5 120 bytes, which the occupancy API does not know: wofdm_lds_workgroups_per_cu)
    if (occ > wofdm_lds_workgroups_per_cu(lds)) occ = wofdm_lds_workgroups_per_cu(lds);
    const int fbuf = wofdm_fbuf_len(g.N, g.T, spw, g.S, g.B);
    HIP_TRY(hipMemcpy(pl->d_geo + WOFDM_G_FBUF, &fbuf, sizeof(int), hipMemcpyHostToDevice));
    const int spwr = wofdm_spwr(spw, g.N, g.S, g.B);
    HIP_TRY(hipMemcpy(pl->d_geo + WOFDM_G_SPWR, &spwr, sizeof(int), hipMemcpyHostToDevice));
    for (int m = 0; m < 4; ++m) pl->fn[m] = fn[m];
    pl->var = var; pl->spw = spw; pl->occ = occ; pl->base.lds_bytes = lds;
    // scale of the on-air samples inside the kernel (wofdm_kparams): 1/N of the IDFT, times the
    // f16 centring of the matrix-pipe layouts
    const bool fm = wofdm_is_firm(spw);
    pl->base.tx_scale = (fm ? pl->firm_sx : 1.0f) / (float)g.N;
    pl->base.dump_unscale_tx = fm ? 1.0f / pl->firm_sx : 1.0f;
    pl->base.dump_unscale_rx = fm ? 1.0f / (pl->firm_sx * pl->firm_sh) : 1.0f;
    return WOFDM_OK;
}

// The gate of launch() below: one mutex and one "last launch" event per device for the whole process.  The two synchronous
// entry points that launch kernels of their own (wofdm_interference, wofdm_tx_psd: kernels WITH swizzled packed arithmetic)
// hold the mutex from their device synchronisation to the end of their kernels, so that no frame launch of another thread
// can slip in beside them.
std::mutex g_gate_mu;
hipEvent_t g_gate_last[64] = {};

int launch(wofdm_plan *pl, int mode, wofdm_kparams &kp, uint64_t total_items, int force_grid,
           hipStream_t stream)
{
    wofdm_kernel_fn fn = pl->fn[mode];
    if (!fn) return fail(WOFDM_E_UNSUPPORTED, "no kernel for n_fft=%d", pl->g.N);
    if (total_items == 0) return WOFDM_OK;
    uint64_t grid = (uint64_t)pl->cus * (uint64_t)pl->occ;
#ifdef WOFDM_DEV_OCC               // developer builds only: workgroups launched per CU (occupancy experiments)
    grid = (uint64_t)pl->cus * (uint64_t)(WOFDM_DEV_OCC);
#endif
    if (grid > total_items) grid = total_items;
    if (force_grid > 0) grid = (uint64_t)force_grid;
    const size_t row = wofdm_noise_scratch_len(pl->g.N, pl->spw);
    if (row && grid * row > pl->nscr_elems) {       // only a forced grid can outgrow the plan's scratch
        HIP_TRY(hipDeviceSynchronize());
        if (pl->d_nscr) (void)hipFree(pl->d_nscr);
        pl->d_nscr = nullptr; pl->nscr_elems = 0;
        HIP_TRY(hipMalloc(&pl->d_nscr, grid * row * sizeof(float2)));
        pl->nscr_elems = grid * row;
    }
    kp.noise_scratch = pl->d_nscr;
    kp.status = pl->d_status;
    kp.dftc = pl->d_dftc;
    kp.rx_scale = pl->d_rxs;
    kp.items_q = total_items / grid;
    kp.items_r = total_items % grid;
    kp.lds_bytes = pl->base.lds_bytes;
    float2 *tm = pl->var == WOFDM_VAR_TXFFT ? (pl->spw == 15 ? reinterpret_cast<float2 *>(pl->d_tspec4) : pl->d_tspec) : pl->d_tmask;
    kp.tx_scale = pl->base.tx_scale;
    kp.dump_unscale_tx = pl->base.dump_unscale_tx;
    kp.dump_unscale_rx = pl->base.dump_unscale_rx;
#ifdef WOFDM_AUDIT
    kp.audit = pl->audit; kp.audit_items = pl->audit_items;
#endif
#ifdef WOFDM_DELAY
    {
        const char *dp = std::getenv("WOFDM_DELAY_POINT"), *dw = std::getenv("WOFDM_DELAY_WAVES"), *dn = std::getenv("WOFDM_DELAY_LEN");
        kp.delay_point = dp ? (uint32_t)std::strtoul(dp, nullptr, 0) : 0;
        kp.delay_waves = dw ? (uint32_t)std::strtoul(dw, nullptr, 0) : 0;
        kp.delay_len = dn ? (uint32_t)std::strtoul(dn, nullptr, 0) : 8;
    }
#endif
    void *args[] = {&kp, &pl->d_wtx, &pl->d_wrx, &pl->d_h, &pl->d_nlin, &pl->d_geo, &pl->d_amask, &tm,
                    &pl->d_fira};
    // One frame kernel at a time per device, whatever streams the callers use: the kernels of layouts 10 / 11 / 12 issue MFMAs
    // in a rhythm that corrupts op_sel-swizzled packed arithmetic of OTHER waves on their SIMDs (wofdm_kernel.hip, mma33);
    // they hold no such instruction themselves, every other kernel of the library does.  Each launch waits for the
    // previous launch of this process on the device (an event; free when it is the same stream) -- a kernel fills the GPU
    // on its own, so nothing is lost.
    // (What the library can NOT order is work it does not launch: kernels of the caller's own -- torch, rocBLAS, RCCL -- on other
    // streams of the device, or another process's.  include/wofdm.h and INTEGRATION.md state the requirement: nothing else runs
    // on the device while a launch of layouts 10 ... 15 is in flight, or the plan is switched to the VALU transforms -- option
    // dft_valu, whose kernels issue one cache-line-aligned chain at a time: hazard 1's safe shape.)
    {
        std::lock_guard<std::mutex> lock(g_gate_mu);
        hipEvent_t *last = g_gate_last;
        const int dev = pl->device & 63;
        if (last[dev]) HIP_TRY(hipStreamWaitEvent(stream, last[dev], 0));
        else HIP_TRY(hipEventCreateWithFlags(&last[dev], hipEventDisableTiming));
        HIP_TRY(hipLaunchKernel(reinterpret_cast<const void *>(fn), dim3((unsigned)grid),
                                dim3(64u * (unsigned)wofdm_waves(pl->spw, pl->g.N, pl->g.S, pl->g.B)), args, kp.lds_bytes, stream));
        HIP_TRY(hipEventRecord(last[dev], stream));
    }
    return WOFDM_OK;
}

// After a synchronisation: has any kernel of this plan reported a problem?
int check_status(wofdm_plan *pl)
{
    unsigned st = 0;
    HIP_TRY(hipMemcpy(&st, pl->d_status, sizeof st, hipMemcpyDeviceToHost));
    if (st != 0)
        return fail(WOFDM_E_HIP, "kernel status 0x%x: a wave timed out waiting for its workgroup "
                    "(results must not be used)", st);
    return WOFDM_OK;
}

int check_launch_args(wofdm_plan *pl, uint64_t frames_per_cell)
{
    if (!pl) return fail(WOFDM_E_INVALID, "plan is NULL");
    if (frames_per_cell > (1ull << 40)) return fail(WOFDM_E_INVALID, "frames_per_cell too large");
    return WOFDM_OK;
}

}  // namespace

extern "C" {

int wofdm_version(void) { return WOFDM_ABI_VERSION; }

const char *wofdm_last_error(void) { return g_err; }

int wofdm_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return fail(WOFDM_E_HIP, "hipGetDeviceCount: %s", hipGetErrorString(e));
    return n;
}

int wofdm_noise_len(const wofdm_cfg *cfg)
{
    geom g;
    int rc = check_cfg(cfg, &g);
    return rc ? rc : g.NL;
}

int wofdm_plan_create(wofdm_plan **out, const wofdm_cfg *cfg, int device, const float *w_tx,
                      const float *w_rx, const float *h, const float *snr_db)
{
    if (!out || !w_tx || !w_rx || !h || !snr_db) return fail(WOFDM_E_INVALID, "NULL argument");
    *out = nullptr;
    geom g;
    int rc = check_cfg(cfg, &g);
    if (rc) return rc;
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev)
        return fail(WOFDM_E_HIP, "device %d not available (%d visible); there is no CPU fallback",
                    device, ndev);
    HIP_TRY(hipSetDevice(device));

    wofdm_plan *pl = new (std::nothrow) wofdm_plan;
    if (!pl) return fail(WOFDM_E_NOMEM, "out of host memory");
    pl->device = device; pl->cfg = *cfg; pl->g = g;
    pl->n_cells = (uint32_t)cfg->n_channels * cfg->n_snr * cfg->n_window_pairs;

    const size_t n_wtx = (size_t)cfg->n_window_pairs * g.P;
    const size_t n_wrx = (size_t)cfg->n_window_pairs * (g.N + g.delta);
    std::vector<float2> hp((size_t)cfg->n_channels * WOFDM_LT, make_float2(0.f, 0.f));
    for (int c = 0; c < cfg->n_channels; ++c)
        for (int l = 0; l < g.L; ++l)
            hp[(size_t)c * WOFDM_LT + l] = make_float2(h[2 * ((size_t)c * g.L + l)],
                                                       h[2 * ((size_t)c * g.L + l) + 1]);
    // Matrix-pipe FIR (wofdm_kernel.hip, phase B of layouts 6 and 7): per channel the 16 x 64 block
    // A[2i + o][2jj + c] = {hr, -hi; hi, hr}[o][c] of tap i + 24 - jj (i < 8 outputs, jj < 32 window
    // samples), scaled by a power of two and split into f16 hi + lo, in the operand layout of
    // v_mfma_f32_16x16x32_f16: lane (m = lane % 16, kg = lane / 16) holds columns 8 kg .. 8 kg + 7
    // of row m of one K = 32 half.  fira[ch][2 half + part][lane] = 8 halves.
    double hmax = 0.0, wmax = 0.0;
    for (float v : std::vector<float>(h, h + 2 * (size_t)cfg->n_channels * g.L)) hmax = std::fmax(hmax, std::fabs((double)v));
    for (size_t i = 0; i < n_wtx; ++i) wmax = std::fmax(wmax, std::fabs((double)w_tx[i]));
    // taps: largest component in [512, 1024); samples: |x| <= wmax * 1.53 (largest 64-QAM point) * N
    // before the table's 1/N, kept below 2^15
    pl->firm_sh = hmax > 0.0 ? (float)std::exp2(9.0 - std::floor(std::log2(hmax))) : 1.0f;
    pl->firm_sx = wmax > 0.0 ? (float)std::exp2(std::floor(std::log2(32768.0 / (wmax * 1.53)))) : 1.0f;
    std::vector<_Float16> fira((size_t)cfg->n_channels * 4 * 64 * 8);
    for (int c = 0; c < cfg->n_channels; ++c)
        for (int a = 0; a < 4; ++a)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 8; ++e) {
                    const int hf = a >> 1, part = a & 1, m = lane & 15, kg = lane >> 4;
                    const int jj = 16 * hf + 4 * kg + (e >> 1), cc = e & 1, i = m >> 1, o = m & 1;
                    const int l = i + 24 - jj;
                    double val = 0.0;
                    if (l >= 0 && l < g.L) {
                        const double hr = h[2 * ((size_t)c * g.L + l)], hi = h[2 * ((size_t)c * g.L + l) + 1];
                        val = (o == 0 ? (cc == 0 ? hr : -hi) : (cc == 0 ? hi : hr)) * (double)pl->firm_sh;
                    }
                    const _Float16 vh = (_Float16)val;
                    const _Float16 vl = (_Float16)(val - (double)vh);
                    fira[(((size_t)c * 4 + a) * 64 + lane) * 8 + e] = part ? vl : vh;
                }
    std::vector<float> nlin(cfg->n_snr);
    for (int i = 0; i < cfg->n_snr; ++i) nlin[i] = (float)std::pow(10.0, -0.1 * (double)snr_db[i]);
    // Matrix-pipe 256-point transforms (wofdm_kernel.hip, mdft_fwd; layouts 10, 11): per lane (a = lane % 16, g = lane / 16)
    // the operand rows of th^(x y), th = exp(-2 pi i / 16), as {Fr, -Fi} (real outputs) / {Fi, Fr} (imaginary outputs) over
    // K = (index, re | im), split into f16 hi + lo:
    //   rows 0..3  stage 1, B operand: column k1 = a, K slot (g, j) <-> n1 = g + 4 j          (re hi, re lo, im hi, im lo)
    //   rows 4..7  stage 2, A operand: row a <-> k2 = a / 4 + 4 (a % 4), K slot (g, j) <-> n2 = 4 g + j
    //   rows 8, 9  the inter-stage twiddles exp(-2 pi i (4 g + j) a / 256), j < 4: real parts, imaginary parts (fp32)
    //   rows 10.. (N = 512, 1024: layout 12) the twiddles in front of the last, radix-N/256 stage, exp(-2 pi i c (lane + 64 j) / N),
    //             c = 1 .. N/256 - 1: real parts, imaginary parts  (N = 256: the same for 1024 points, the Tx mask of layout 15)
    // (rows beyond the first ten: N >= 512, N = 128, and N = 256 for the 1024-point transforms of the Tx mask -- layout 15)
    const int dft_nc = g.N >= 512 ? g.N / 256 : (g.N == 128 ? 2 : (g.N == 256 ? 4 : 1));
    const int dft_n2 = g.N == 256 ? 1024 : g.N;
    std::vector<uint32_t> dftc((size_t)(10 + 2 * (dft_nc - 1)) * 64 * 4);
    {
        const double PI = 3.14159265358979323846;
        for (int lane = 0; lane < 64; ++lane) {
            const int a = lane & 15, gq = lane >> 4;
            for (int j = 0; j < 4; ++j) {
                for (int st = 0; st < 2; ++st) {
                    const int x = st == 0 ? gq + 4 * j : 4 * gq + j;         // contracted index of the K slot
                    const int y = st == 0 ? a : (a >> 2) + 4 * (a & 3);      // output index of the lane
                    const double ang = -2.0 * PI * (double)((x * y) & 15) / 16.0;
                    double fr = std::cos(ang), fi = std::sin(ang);
                    if (((x * y) & 3) == 0) {                                 // multiples of a quarter turn: exact 0, +-1
                        fr = std::round(fr); fi = std::round(fi);
                    }
                    const double vals[2][2] = {{fr, -fi}, {fi, fr}};          // [re | im output][K = re | im input]
                    for (int o = 0; o < 2; ++o) {
                        uint32_t whi = 0, wlo = 0;
                        for (int c = 0; c < 2; ++c) {
                            const _Float16 vh = (_Float16)vals[o][c];
                            const _Float16 vl = (_Float16)(vals[o][c] - (double)vh);
                            uint16_t bh, bl;
                            std::memcpy(&bh, &vh, 2); std::memcpy(&bl, &vl, 2);
                            whi |= (uint32_t)bh << (16 * c);
                            wlo |= (uint32_t)bl << (16 * c);
                        }
                        dftc[((size_t)(4 * st + 2 * o + 0) * 64 + lane) * 4 + j] = whi;
                        dftc[((size_t)(4 * st + 2 * o + 1) * 64 + lane) * 4 + j] = wlo;
                    }
                }
                double ang = -2.0 * PI * (double)((4 * gq + j) * a) / 256.0;
                // (N = 64, 128 -- layouts 13 / 14: ONE 16-point stage; rows 8, 9 (and 10, 11 at N = 128) hold the twiddles in front
                // of the radix-N/16 stage, exp(-2 pi i (4 p + j) (lane % 16) / N), p = part of the symbol's N/16 stage-1 outputs)
                if (g.N <= 128) ang = -2.0 * PI * (double)(j * a) / (double)g.N;
                const float twr = (float)std::cos(ang), twi = (float)std::sin(ang);
                std::memcpy(&dftc[((size_t)8 * 64 + lane) * 4 + j], &twr, 4);
                std::memcpy(&dftc[((size_t)9 * 64 + lane) * 4 + j], &twi, 4);
                if (g.N == 128) {
                    const double a3 = -2.0 * PI * (double)((4 + j) * a) / 128.0;
                    const float t3r = (float)std::cos(a3), t3i = (float)std::sin(a3);
                    std::memcpy(&dftc[((size_t)10 * 64 + lane) * 4 + j], &t3r, 4);
                    std::memcpy(&dftc[((size_t)11 * 64 + lane) * 4 + j], &t3i, 4);
                }
                for (int c = 1; g.N >= 256 && c < dft_nc; ++c) {
                    const double a2 = -2.0 * PI * (double)((c * (lane + 64 * j)) % dft_n2) / (double)dft_n2;
                    const float t2r = (float)std::cos(a2), t2i = (float)std::sin(a2);
                    std::memcpy(&dftc[((size_t)(10 + 2 * (c - 1)) * 64 + lane) * 4 + j], &t2r, 4);
                    std::memcpy(&dftc[((size_t)(11 + 2 * (c - 1)) * 64 + lane) * 4 + j], &t2i, 4);
                }
            }
        }
    }
    // received samples in kernel units: natural units (unit-power symbols, 1/N in the IDFT: rms 1/sqrt(N)) x firm_sx x firm_sh x
    // |h|, plus noise; a power of two per (snr, channel) brings their rms to about 16 (everything behind is homogeneous
    // in it: the equaliser divides by the pilot)
    std::vector<float> rxs((size_t)cfg->n_snr * cfg->n_channels, 1.0f);
    for (int i = 0; i < cfg->n_snr; ++i)
        for (int c = 0; c < cfg->n_channels; ++c) {
            double e = 0.0;
            for (int l = 0; l < g.L; ++l) {
                const double hr = h[2 * ((size_t)c * g.L + l)], hi = h[2 * ((size_t)c * g.L + l) + 1];
                e += hr * hr + hi * hi;
            }
            const double rms = (double)pl->firm_sx * (double)pl->firm_sh * std::sqrt(e * (1.0 + (double)nlin[i]) / (double)g.N)
                               * std::fmax(wmax, 1e-30);
            if (rms > 0.0 && std::isfinite(rms))
                rxs[(size_t)i * cfg->n_channels + c] = (float)std::exp2(std::round(4.0 - std::log2(rms)));
        }

#define PLAN_TRY(expr)                                                                        \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) {                                                               \
            wofdm_plan_destroy(pl);                                                           \
            return fail(e_ == hipErrorOutOfMemory ? WOFDM_E_NOMEM : WOFDM_E_HIP, "%s: %s",    \
                        #expr, hipGetErrorString(e_));                                        \
        }                                                                                     \
    } while (0)

    PLAN_TRY(hipMalloc(&pl->d_wtx, n_wtx * sizeof(float)));
    PLAN_TRY(hipMalloc(&pl->d_wrx, n_wrx * sizeof(float)));
    PLAN_TRY(hipMalloc(&pl->d_h, hp.size() * sizeof(float2)));
    PLAN_TRY(hipMalloc(&pl->d_nlin, nlin.size() * sizeof(float)));
    PLAN_TRY(hipMemcpy(pl->d_wtx, w_tx, n_wtx * sizeof(float), hipMemcpyHostToDevice));
    PLAN_TRY(hipMemcpy(pl->d_wrx, w_rx, n_wrx * sizeof(float), hipMemcpyHostToDevice));
    PLAN_TRY(hipMemcpy(pl->d_h, hp.data(), hp.size() * sizeof(float2), hipMemcpyHostToDevice));
    PLAN_TRY(hipMemcpy(pl->d_nlin, nlin.data(), nlin.size() * sizeof(float), hipMemcpyHostToDevice));
    PLAN_TRY(hipMalloc(&pl->d_fira, fira.size() * sizeof(_Float16)));
    PLAN_TRY(hipMemcpy(pl->d_fira, fira.data(), fira.size() * sizeof(_Float16), hipMemcpyHostToDevice));
    PLAN_TRY(hipMalloc(&pl->d_dftc, dftc.size() * sizeof(uint32_t)));
    PLAN_TRY(hipMemcpy(pl->d_dftc, dftc.data(), dftc.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    PLAN_TRY(hipMalloc(&pl->d_rxs, rxs.size() * sizeof(float)));
    PLAN_TRY(hipMemcpy(pl->d_rxs, rxs.data(), rxs.size() * sizeof(float), hipMemcpyHostToDevice));
    PLAN_TRY(hipEventCreate(&pl->ev0));
    PLAN_TRY(hipEventCreate(&pl->ev1));

    int geo[WOFDM_G_COUNT];
    geo[WOFDM_G_S] = g.S; geo[WOFDM_G_MU] = g.mu; geo[WOFDM_G_RHO] = g.rho; geo[WOFDM_G_BETA] = g.beta;
    geo[WOFDM_G_DELTA] = g.delta; geo[WOFDM_G_GAMMA] = g.gamma; geo[WOFDM_G_KAPPA] = g.kappa;
    geo[WOFDM_G_L] = g.L; geo[WOFDM_G_P] = g.P; geo[WOFDM_G_B] = g.B; geo[WOFDM_G_T] = g.T;
    geo[WOFDM_G_NL] = g.NL; geo[WOFDM_G_NSNR] = cfg->n_snr; geo[WOFDM_G_NCH] = cfg->n_channels;
    geo[WOFDM_G_FBUF] = 0;                       // set by configure()
    geo[WOFDM_G_NACT] = g.N;
    geo[WOFDM_G_SPWR] = 0;                       // set by configure()
    PLAN_TRY(hipMalloc(&pl->d_geo, sizeof geo));
    PLAN_TRY(hipMemcpy(pl->d_geo, geo, sizeof geo, hipMemcpyHostToDevice));

    wofdm_kparams &kp = pl->base;
    kp.n_cells = pl->n_cells; kp.first_cell = 0; kp.inject_base_cell = 0;
    kp.seed_lo = (uint32_t)cfg->seed; kp.seed_hi = (uint32_t)(cfg->seed >> 32);

    hipDeviceProp_t prop;
    PLAN_TRY(hipGetDeviceProperties(&prop, device));
    pl->cus = prop.multiProcessorCount;
    PLAN_TRY(hipMalloc(&pl->d_status, sizeof(unsigned)));
    PLAN_TRY(hipMemset(pl->d_status, 0, sizeof(unsigned)));
    if ((rc = configure(pl)) != WOFDM_OK) {
        wofdm_plan_destroy(pl);
        return rc;
    }
    const int occ = pl->occ;
    if (const size_t row = wofdm_noise_scratch_len(g.N, pl->spw)) {
        pl->nscr_elems = (uint64_t)pl->cus * (uint64_t)occ * row;
        PLAN_TRY(hipMalloc(&pl->d_nscr, pl->nscr_elems * sizeof(float2)));
        PLAN_TRY(hipMemset(pl->d_nscr, 0, pl->nscr_elems * sizeof(float2)));
        PLAN_TRY(hipDeviceSynchronize());
    }
#undef PLAN_TRY
    *out = pl;
    return WOFDM_OK;
}

int wofdm_plan_destroy(wofdm_plan *pl)
{
    if (!pl) return WOFDM_OK;
    (void)hipSetDevice(pl->device);
    if (pl->d_wtx) (void)hipFree(pl->d_wtx);
    if (pl->d_wrx) (void)hipFree(pl->d_wrx);
    if (pl->d_h) (void)hipFree(pl->d_h);
    if (pl->d_nlin) (void)hipFree(pl->d_nlin);
    if (pl->d_geo) (void)hipFree(pl->d_geo);
    if (pl->d_nscr) (void)hipFree(pl->d_nscr);
    if (pl->d_amask) (void)hipFree(pl->d_amask);
    if (pl->d_tmask) (void)hipFree(pl->d_tmask);
    if (pl->d_tspec) (void)hipFree(pl->d_tspec);
    if (pl->d_tspec4) (void)hipFree(pl->d_tspec4);
    if (pl->d_status) (void)hipFree(pl->d_status);
    if (pl->d_fira) (void)hipFree(pl->d_fira);
    if (pl->d_dftc) (void)hipFree(pl->d_dftc);
    if (pl->d_rxs) (void)hipFree(pl->d_rxs);
    if (pl->ev0) (void)hipEventDestroy(pl->ev0);
    if (pl->ev1) (void)hipEventDestroy(pl->ev1);
    delete pl;
    return WOFDM_OK;
}

int wofdm_plan_set_allocation(wofdm_plan *pl, const uint8_t *active)
{
    if (!pl) return fail(WOFDM_E_INVALID, "plan is NULL");
    HIP_TRY(hipSetDevice(pl->device));
    HIP_TRY(hipDeviceSynchronize());           // no launch of this plan may still read the mask
    const int N = pl->g.N, NQ = N / 4;
    const size_t amask_words = wofdm_amask_words(N);
    int nact = N;
    if (active) {
        nact = 0;
        // [0, NQ): word j, byte r <-> subcarrier j + r NQ; [NQ, 2 NQ): the quarter-wave order of
        // the N = 256 kernels, word 16 q + l, byte r <-> subcarrier l + 16 (q + 4 r)
        // N >= 512, [NQ, NQ + 256): the input element order of layout 12, word lane + 64 j, byte c <-> subcarrier
        // N/16 (lane / 16 + 4 j) + NC (lane % 16) + c, NC = N / 256; N = 256: the same with NC = 1 (layout 15) in [2 NQ, 2 NQ + 256)
        std::vector<uint32_t> words(amask_words, 0u);
        for (int n = 0; n < N; ++n) {
            if (active[n]) { ++nact; continue; }
            words[(size_t)(n % NQ)] |= 0x80u << (8 * (n / NQ));
            if (N == 256) {
                const int l = n & 15, t = n >> 4;
                words[(size_t)(NQ + 16 * (t & 3) + l)] |= 0x80u << (8 * (t >> 2));
            }
            if (N <= 128) {
                // [NQ, NQ + 256): layouts 13 / 14, word 64 t + lane (set t), byte j <-> subcarrier N/16 (lane / 16 + 4 j) + c,
                // c = 4 (t % (N / 64)) + lane % 4
                const int sc = N / 16, scs = sc / 4;
                for (int t = 0; t < 4; ++t)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j = 0; j < 4; ++j)
                            if (sc * (lane / 16 + 4 * j) + 4 * (t % scs) + (lane & 3) == n)
                                words[(size_t)(NQ + 64 * t + lane)] |= 0x80u << (8 * j);
            }
            if (N >= 512 || N == 256) {
                // (N = 256, layout 15: one set, behind the quarter-wave part)
                const int nc = N >= 512 ? N / 256 : 1, a = n / (N / 16), r = n % (N / 16), b = r / nc, c = r % nc;   // n = N/16 a + NC b + c
                const int lane = 16 * (a & 3) + b, j = a >> 2;                                         // a = g + 4 j
                words[(size_t)((N == 256 ? 2 * NQ : NQ) + lane + 64 * j)] |= 0x80u << (8 * c);
            }
        }
        if (nact == 0) return fail(WOFDM_E_INVALID, "allocation loads no subcarrier");
        if (!pl->d_amask) HIP_TRY(hipMalloc(&pl->d_amask, amask_words * sizeof(uint32_t)));
        HIP_TRY(hipMemcpy(pl->d_amask, words.data(), amask_words * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    if (!active && pl->d_amask) HIP_TRY(hipMemset(pl->d_amask, 0, amask_words * sizeof(uint32_t)));
    HIP_TRY(hipMemcpy(pl->d_geo + WOFDM_G_NACT, &nact, sizeof(int), hipMemcpyHostToDevice));
    pl->has_alloc = active && nact < N;
    return configure(pl);
}

int wofdm_plan_set_tx_mask(wofdm_plan *pl, const float *mask)
{
    if (!pl) return fail(WOFDM_E_INVALID, "plan is NULL");
    HIP_TRY(hipSetDevice(pl->device));
    HIP_TRY(hipDeviceSynchronize());
    if (!mask) {
        pl->has_mask = false;
        return configure(pl);
    }
    // impulse response of the mask: g = IDFT_{2P-1}(mask), in double on the host (complex in
    // general: main_channel_mask.m's centred raised cosine is not even around bin 0 when P is odd)
    const int Lm = 2 * pl->g.P - 1;
    std::vector<float2> gtd((size_t)Lm);
    for (int n = 0; n < Lm; ++n) {
        double re = 0.0, im = 0.0;
        for (int k = 0; k < Lm; ++k) {
            const double a = 2.0 * M_PI * (double)(((long long)k * n) % Lm) / (double)Lm;
            re += (double)mask[k] * std::cos(a);
            im += (double)mask[k] * std::sin(a);
        }
        gtd[(size_t)n] = make_float2((float)(re / Lm), (float)(im / Lm));
    }
    if (!pl->d_tmask) HIP_TRY(hipMalloc(&pl->d_tmask, (size_t)Lm * sizeof(float2)));
    HIP_TRY(hipMemcpy(pl->d_tmask, gtd.data(), (size_t)Lm * sizeof(float2), hipMemcpyHostToDevice));
    {
        // fast-convolution spectrum: FFT_MF of gt[t] = g[(t - (P-1)) mod (2P-1)], t < 3P-2, with the
        // 1/MF of the inverse transform folded in (wofdm_kernel.hip, TXFFT)
        const int MF = WOFDM_TXFFT_LEN, P = pl->g.P, G = 3 * P - 2;
        std::vector<float2> spec((size_t)MF, make_float2(0.f, 0.f));
        if (G <= MF) {
            std::vector<double> gr((size_t)G), gi((size_t)G), cw((size_t)MF), sw((size_t)MF);
            for (int t = 0; t < G; ++t) {
                const int kk = ((t - (P - 1)) % Lm + Lm) % Lm;
                double re = 0.0, im = 0.0;          // g in double again (gtd is rounded to float)
                for (int k = 0; k < Lm; ++k) {
                    const double a = 2.0 * M_PI * (double)(((long long)k * kk) % Lm) / (double)Lm;
                    re += (double)mask[k] * std::cos(a);
                    im += (double)mask[k] * std::sin(a);
                }
                gr[(size_t)t] = re / Lm; gi[(size_t)t] = im / Lm;
            }
            for (int q = 0; q < MF; ++q) {
                cw[(size_t)q] = std::cos(2.0 * M_PI * q / MF); sw[(size_t)q] = -std::sin(2.0 * M_PI * q / MF);
            }
            for (int f = 0; f < MF; ++f) {
                double re = 0.0, im = 0.0;
                for (int t = 0; t < G; ++t) {
                    const int q = (int)(((long long)f * t) % MF);
                    re += gr[(size_t)t] * cw[(size_t)q] - gi[(size_t)t] * sw[(size_t)q];
                    im += gr[(size_t)t] * sw[(size_t)q] + gi[(size_t)t] * cw[(size_t)q];
                }
                spec[(size_t)f] = make_float2((float)(re / MF), (float)(im / MF));
            }
        }
        if (!pl->d_tspec) HIP_TRY(hipMalloc(&pl->d_tspec, (size_t)MF * sizeof(float2)));
        HIP_TRY(hipMemcpy(pl->d_tspec, spec.data(), (size_t)MF * sizeof(float2), hipMemcpyHostToDevice));
        if (MF == 1024) {
            std::vector<float> spec4((size_t)2 * MF);
            for (int kc = 0; kc < 4; ++kc)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 4; ++j) {
                        const float2 v = spec[(size_t)(lane + 64 * j + 256 * kc)];
                        spec4[((size_t)(2 * kc + 0) * 64 + lane) * 4 + j] = 16.0f * v.x;
                        spec4[((size_t)(2 * kc + 1) * 64 + lane) * 4 + j] = 16.0f * v.y;
                    }
            if (!pl->d_tspec4) HIP_TRY(hipMalloc(&pl->d_tspec4, spec4.size() * sizeof(float)));
            HIP_TRY(hipMemcpy(pl->d_tspec4, spec4.data(), spec4.size() * sizeof(float), hipMemcpyHostToDevice));
        }
    }
    if (!pl->d_amask) {        // the mask kernels always read an allocation word
        const size_t amask_words = wofdm_amask_words(pl->g.N);
        HIP_TRY(hipMalloc(&pl->d_amask, amask_words * sizeof(uint32_t)));
        HIP_TRY(hipMemset(pl->d_amask, 0, amask_words * sizeof(uint32_t)));
    }
    const bool had = pl->has_mask;
    pl->has_mask = true;
    const int rc = configure(pl);
    if (rc != WOFDM_OK) pl->has_mask = had;
    return rc;
}

int wofdm_plan_set_option(wofdm_plan *pl, int32_t option, int32_t value)
{
    if (!pl) return fail(WOFDM_E_INVALID, "plan is NULL");
    HIP_TRY(hipSetDevice(pl->device));
    HIP_TRY(hipDeviceSynchronize());           // no launch of this plan may still be running on the old choice
    const bool direct = pl->force_direct_mask, valu = pl->fir_valu, dvalu = pl->dft_valu;
    const int cap = pl->max_spw;
    switch (option) {
    case WOFDM_OPT_FIR_VALU: pl->fir_valu = value != 0; break;
    case WOFDM_OPT_MAX_SPW:
        if (value != 0 && value != 1 && value != 2 && value != 4)
            return fail(WOFDM_E_INVALID, "WOFDM_OPT_MAX_SPW takes 0, 1, 2 or 4");
        pl->max_spw = value;
        break;
    case WOFDM_OPT_TXMASK_DIRECT: pl->force_direct_mask = value != 0; break;
    case WOFDM_OPT_DFT_VALU: pl->dft_valu = value != 0; break;
    default: return fail(WOFDM_E_INVALID, "unknown option %d", (int)option);
    }
    const int rc = configure(pl);
    if (rc != WOFDM_OK) { pl->force_direct_mask = direct; pl->fir_valu = valu; pl->max_spw = cap; pl->dft_valu = dvalu; }
    // (the noise scratch rows are sized per layout: a forced grid in launch() regrows them, a new layout here)
    if (rc == WOFDM_OK) {
        const size_t row = wofdm_noise_scratch_len(pl->g.N, pl->spw);
        const uint64_t wgs = (uint64_t)pl->cus * (uint64_t)pl->occ;
        if (row && wgs * row > pl->nscr_elems) {
            if (pl->d_nscr) (void)hipFree(pl->d_nscr);
            pl->d_nscr = nullptr; pl->nscr_elems = 0;
            HIP_TRY(hipMalloc(&pl->d_nscr, wgs * row * sizeof(float2)));
            HIP_TRY(hipMemset(pl->d_nscr, 0, wgs * row * sizeof(float2)));
            pl->nscr_elems = wgs * row;
        }
    }
    return rc;
}

int wofdm_plan_status(wofdm_plan *pl)
{
    if (!pl) return fail(WOFDM_E_INVALID, "plan is NULL");
    HIP_TRY(hipSetDevice(pl->device));
    return check_status(pl);
}

int wofdm_plan_info(wofdm_plan *pl, int32_t info[5])
{
    if (!pl || !info) return fail(WOFDM_E_INVALID, "NULL argument");
    info[0] = wofdm_waves(pl->spw, pl->g.N, pl->g.S, pl->g.B);
    info[1] = (int32_t)pl->base.lds_bytes;
    info[2] = pl->cus * pl->occ;
    info[3] = pl->occ;
    info[4] = pl->cus;
    return WOFDM_OK;
}

#ifdef WOFDM_AUDIT
// developer build only: device buffer of [grid][items][16][8] words for the per-frame records
extern "C" int wofdm_plan_set_audit(wofdm_plan *pl, void *dev, uint32_t items)
{
    if (!pl) return WOFDM_E_INVALID;
    pl->audit = static_cast<uint32_t *>(dev); pl->audit_items = items;
    return WOFDM_OK;
}
#endif
int wofdm_plan_kernel_id(wofdm_plan *pl, int32_t id[2])
{
    if (!pl || !id) return fail(WOFDM_E_INVALID, "NULL argument");
    id[0] = pl->spw;
    id[1] = pl->var;
    return WOFDM_OK;
}

int wofdm_plan_launch(wofdm_plan *pl, uint64_t frame_offset, uint64_t frames_per_cell,
                      uint64_t *counts_dev, void *stream)
{
    int rc = check_launch_args(pl, frames_per_cell);
    if (rc) return rc;
    if (!counts_dev) return fail(WOFDM_E_INVALID, "counts_dev is NULL");
    HIP_TRY(hipSetDevice(pl->device));
    wofdm_kparams kp = pl->base;
    kp.frames_per_cell = frames_per_cell; kp.frame_offset = frame_offset;
    kp.counts = reinterpret_cast<unsigned long long *>(counts_dev);
    return launch(pl, WOFDM_MODE_GEN, kp, frames_per_cell * pl->n_cells, 0,
                  static_cast<hipStream_t>(stream));
}

int wofdm_plan_launch_timed(wofdm_plan *pl, uint64_t frame_offset, uint64_t frames_per_cell,
                            uint64_t *counts_dev, void *stream, float *kernel_ms)
{
    if (!pl || !kernel_ms) return fail(WOFDM_E_INVALID, "NULL argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    HIP_TRY(hipSetDevice(pl->device));
    HIP_TRY(hipEventRecord(pl->ev0, st));
    int rc = wofdm_plan_launch(pl, frame_offset, frames_per_cell, counts_dev, stream);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(pl->ev1, st));
    HIP_TRY(hipEventSynchronize(pl->ev1));
    HIP_TRY(hipEventElapsedTime(kernel_ms, pl->ev0, pl->ev1));
    return check_status(pl);
}

int wofdm_plan_launch_injected(wofdm_plan *pl, uint64_t frames_per_cell, const uint8_t *labels_dev,
                               const float *unit_noise_dev, uint64_t *counts_dev, void *stream)
{
    int rc = check_launch_args(pl, frames_per_cell);
    if (rc) return rc;
    if (!counts_dev || !labels_dev || !unit_noise_dev) return fail(WOFDM_E_INVALID, "NULL argument");
    HIP_TRY(hipSetDevice(pl->device));
    wofdm_kparams kp = pl->base;
    kp.frames_per_cell = frames_per_cell; kp.frame_offset = 0;
    kp.counts = reinterpret_cast<unsigned long long *>(counts_dev);
    kp.labels = labels_dev;
    kp.unit_noise = reinterpret_cast<const float2 *>(unit_noise_dev);
    return launch(pl, WOFDM_MODE_INJECT, kp, frames_per_cell * pl->n_cells, 0,
                  static_cast<hipStream_t>(stream));
}

int wofdm_plan_dump_frame(wofdm_plan *pl, uint32_t cell, uint64_t frame, const uint8_t *labels,
                          const float *unit_noise, uint64_t *counts, wofdm_dump *out)
{
    if (!pl || !counts || !out) return fail(WOFDM_E_INVALID, "NULL argument");
    if (cell >= pl->n_cells) return fail(WOFDM_E_INVALID, "cell %u out of range", cell);
    if ((labels == nullptr) != (unit_noise == nullptr))
        return fail(WOFDM_E_INVALID, "inject both labels and unit_noise or neither");
    const geom &g = pl->g;
    HIP_TRY(hipSetDevice(pl->device));
    const size_t SN = (size_t)g.S * g.N, CL = (size_t)g.T + g.L - 1, SB = (size_t)g.S * g.B;
    // one device arena: [counts 4xu64][X][tx][conv][rx][Y][Xhat][noise] float2, gain, labels
    const size_t n_f2 = SN + g.T + CL + SB + SN + SN + (size_t)g.NL;
    const size_t bytes = 32 + n_f2 * sizeof(float2) + 16 + 2 * SN + (labels ? SN + (size_t)g.NL * 8 : 0) + 64;
    unsigned char *arena = nullptr;
    HIP_TRY(hipMalloc(&arena, bytes));
    int rc = WOFDM_OK;
    do {
        if (hipMemset(arena, 0, bytes) != hipSuccess) { rc = fail(WOFDM_E_HIP, "hipMemset failed"); break; }
        wofdm_kparams kp = pl->base;
        unsigned char *ptr = arena;
        kp.counts = reinterpret_cast<unsigned long long *>(ptr); ptr += 32;    // entry 0 = `cell` (inject_base_cell below)
        float2 *f2 = reinterpret_cast<float2 *>(ptr);
        kp.dump.X = f2; f2 += SN;
        kp.dump.tx = f2; f2 += g.T;
        kp.dump.conv = f2; f2 += CL;
        kp.dump.rx = f2; f2 += SB;
        kp.dump.Y = f2; f2 += SN;
        kp.dump.Xhat = f2; f2 += SN;
        kp.dump.unit_noise = f2; f2 += g.NL;
        kp.dump.gain = reinterpret_cast<float *>(f2);
        kp.dump.sink = f2 + 1;                     // inside the 16-byte pad behind the gain
        unsigned char *b = reinterpret_cast<unsigned char *>(f2) + 16;
        kp.dump.labels_tx = b; b += SN;
        kp.dump.labels_rx = b; b += SN;
        if (labels) {
            uint8_t *dl = b; b += SN;
            b = reinterpret_cast<unsigned char *>(((uintptr_t)b + 15) & ~(uintptr_t)15);
            float2 *dn = reinterpret_cast<float2 *>(b);
            if (hipMemcpy(dl, labels, SN, hipMemcpyHostToDevice) != hipSuccess ||
                hipMemcpy(dn, unit_noise, (size_t)g.NL * 8, hipMemcpyHostToDevice) != hipSuccess) {
                rc = fail(WOFDM_E_HIP, "hipMemcpy of injected inputs failed"); break;
            }
            kp.labels = dl; kp.unit_noise = dn;
        }
        kp.first_cell = cell; kp.n_cells = 1; kp.inject_base_cell = cell;
        kp.frames_per_cell = 1; kp.frame_offset = frame;
        rc = launch(pl, labels ? WOFDM_MODE_DUMP_INJECT : WOFDM_MODE_DUMP_GEN, kp, 1, 1, nullptr);
        if (rc) break;
        if (hipDeviceSynchronize() != hipSuccess) { rc = fail(WOFDM_E_HIP, "dump kernel failed: %s", hipGetErrorString(hipGetLastError())); break; }
        uint64_t c4[4];
        auto down = [&](void *dst, const void *src, size_t n) {
            return !dst || hipMemcpy(dst, src, n, hipMemcpyDeviceToHost) == hipSuccess;
        };
        bool ok = down(c4, arena, 32) && down(out->X, kp.dump.X, SN * 8) &&
                  down(out->tx, kp.dump.tx, (size_t)g.T * 8) && down(out->conv, kp.dump.conv, CL * 8) &&
                  down(out->rx, kp.dump.rx, SB * 8) && down(out->Y, kp.dump.Y, SN * 8) &&
                  down(out->Xhat, kp.dump.Xhat, (SN - g.N) * 8) &&
                  down(out->unit_noise, kp.dump.unit_noise, (size_t)g.NL * 8) &&
                  down(out->gain, kp.dump.gain, 4) && down(out->labels_tx, kp.dump.labels_tx, SN) &&
                  down(out->labels_rx, kp.dump.labels_rx, SN - g.N);
        if (!ok) { rc = fail(WOFDM_E_HIP, "hipMemcpy of dump failed"); break; }
        // The kernels (instrumented and production alike) leave the circular shift of the Rx chain (m:313-333) out:
        // their Y is the reference's times e^{-2 pi i (kappa + delta/2) n / N} on every symbol, which the one-tap
        // equaliser divides out.  The dump hands the reference's Y back: the ramp is undone here, on the host.
        if (out->Y && (g.kappa + g.delta / 2) % g.N != 0) {
            const int c = (g.kappa + g.delta / 2) % g.N;
            for (int n = 0; n < g.N; ++n) {
                const double a = 2.0 * M_PI * (double)(((long long)c * n) % g.N) / (double)g.N;
                const float cr = (float)std::cos(a), ci = (float)std::sin(a);
                for (int sy = 0; sy < g.S; ++sy) {
                    float *y = out->Y + 2 * ((size_t)sy * g.N + n);
                    const float re = y[0] * cr - y[1] * ci, im = y[0] * ci + y[1] * cr;
                    y[0] = re; y[1] = im;
                }
            }
        }
        for (int i = 0; i < 4; ++i) counts[i] += c4[i];
    } while (0);
    (void)hipFree(arena);
    return rc;
}

int wofdm_run(const wofdm_cfg *cfg, int device, const float *w_tx, const float *w_rx,
              const float *h, const float *snr_db, uint64_t *counts)
{
    if (!counts) return fail(WOFDM_E_INVALID, "counts is NULL");
    wofdm_plan *pl = nullptr;
    int rc = wofdm_plan_create(&pl, cfg, device, w_tx, w_rx, h, snr_db);
    if (rc) return rc;
    const size_t n = (size_t)pl->n_cells * 4;
    uint64_t *d = nullptr;
    std::vector<uint64_t> hc(n);
    do {
        if (hipMalloc(&d, n * 8) != hipSuccess || hipMemset(d, 0, n * 8) != hipSuccess) {
            rc = fail(WOFDM_E_NOMEM, "counter allocation failed"); break;
        }
        rc = wofdm_plan_launch(pl, cfg->frame_offset, cfg->frames_per_cell, d, nullptr);
        if (rc) break;
        if (hipDeviceSynchronize() != hipSuccess ||
            hipMemcpy(hc.data(), d, n * 8, hipMemcpyDeviceToHost) != hipSuccess) {
            rc = fail(WOFDM_E_HIP, "kernel or copy-back failed: %s", hipGetErrorString(hipGetLastError()));
            break;
        }
        if ((rc = check_status(pl)) != WOFDM_OK) break;
        for (size_t i = 0; i < n; ++i) counts[i] += hc[i];
    } while (0);
    if (d) (void)hipFree(d);
    wofdm_plan_destroy(pl);
    return rc;
}

int wofdm_run_injected(const wofdm_cfg *cfg, int device, const float *w_tx, const float *w_rx,
                       const float *h, const float *snr_db, const uint8_t *labels,
                       const float *unit_noise, uint64_t *counts)
{
    if (!counts || !labels || !unit_noise) return fail(WOFDM_E_INVALID, "NULL argument");
    wofdm_plan *pl = nullptr;
    int rc = wofdm_plan_create(&pl, cfg, device, w_tx, w_rx, h, snr_db);
    if (rc) return rc;
    const size_t n = (size_t)pl->n_cells * 4;
    const size_t frames = (size_t)pl->n_cells * cfg->frames_per_cell;
    const size_t lb = frames * pl->g.S * pl->g.N, nb = frames * (size_t)pl->g.NL * 8;
    uint64_t *d = nullptr;
    uint8_t *dl = nullptr;
    float *dn = nullptr;
    std::vector<uint64_t> hc(n);
    do {
        if (hipMalloc(&d, n * 8) != hipSuccess || hipMemset(d, 0, n * 8) != hipSuccess ||
            hipMalloc(&dl, lb ? lb : 1) != hipSuccess || hipMalloc(&dn, nb ? nb : 8) != hipSuccess) {
            rc = fail(WOFDM_E_NOMEM, "device allocation failed"); break;
        }
        if (hipMemcpy(dl, labels, lb, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(dn, unit_noise, nb, hipMemcpyHostToDevice) != hipSuccess) {
            rc = fail(WOFDM_E_HIP, "upload failed"); break;
        }
        rc = wofdm_plan_launch_injected(pl, cfg->frames_per_cell, dl, dn, d, nullptr);
        if (rc) break;
        if (hipDeviceSynchronize() != hipSuccess ||
            hipMemcpy(hc.data(), d, n * 8, hipMemcpyDeviceToHost) != hipSuccess) {
            rc = fail(WOFDM_E_HIP, "kernel or copy-back failed: %s", hipGetErrorString(hipGetLastError()));
            break;
        }
        if ((rc = check_status(pl)) != WOFDM_OK) break;
        for (size_t i = 0; i < n; ++i) counts[i] += hc[i];
    } while (0);
    if (d) (void)hipFree(d);
    if (dl) (void)hipFree(dl);
    if (dn) (void)hipFree(dn);
    wofdm_plan_destroy(pl);
    return rc;
}

int wofdm_interference(const wofdm_cfg *cfg, int device, const float *w_tx, const float *w_rx,
                       const float *h, float *power)
{
    if (!w_tx || !w_rx || !h || !power) return fail(WOFDM_E_INVALID, "NULL argument");
    geom g;
    int rc = check_cfg(cfg, &g);
    if (rc) return rc;
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev)
        return fail(WOFDM_E_HIP, "device %d not available (%d visible); there is no CPU fallback", device, ndev);
    HIP_TRY(hipSetDevice(device));
    const int jobs = cfg->n_window_pairs * cfg->n_channels;
    const size_t n_wtx = (size_t)cfg->n_window_pairs * g.P, n_wrx = (size_t)cfg->n_window_pairs * (g.N + g.delta);
    std::vector<float2> hp((size_t)cfg->n_channels * WOFDM_LT, make_float2(0.f, 0.f));
    for (int c = 0; c < cfg->n_channels; ++c)
        for (int l = 0; l < g.L; ++l)
            hp[(size_t)c * WOFDM_LT + l] = make_float2(h[2 * ((size_t)c * g.L + l)], h[2 * ((size_t)c * g.L + l) + 1]);
    float *d_wtx = nullptr, *d_wrx = nullptr, *d_pow = nullptr;
    float2 *d_h = nullptr;
    rc = WOFDM_OK;
    do {
        if (hipMalloc(&d_wtx, n_wtx * 4) != hipSuccess || hipMalloc(&d_wrx, n_wrx * 4) != hipSuccess ||
            hipMalloc(&d_h, hp.size() * 8) != hipSuccess || hipMalloc(&d_pow, (size_t)jobs * g.N * 4) != hipSuccess) {
            rc = fail(WOFDM_E_NOMEM, "device allocation failed"); break;
        }
        if (hipMemcpy(d_wtx, w_tx, n_wtx * 4, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(d_wrx, w_rx, n_wrx * 4, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(d_h, hp.data(), hp.size() * 8, hipMemcpyHostToDevice) != hipSuccess) {
            rc = fail(WOFDM_E_HIP, "upload failed"); break;
        }
        // (no frame kernel of this process beside these kernels: the gate of launch() is held until they have finished)
        std::lock_guard<std::mutex> gate(g_gate_mu);
        (void)hipDeviceSynchronize();
        hipError_t e = hipErrorInvalidValue;
        if (g.N == 64) e = wofdm_interf_launch_n64(jobs, g.P, g.B, g.mu, g.delta, g.gamma, g.kappa, cfg->n_channels, d_wtx, d_wrx, d_h, d_pow, nullptr);
        if (g.N == 128) e = wofdm_interf_launch_n128(jobs, g.P, g.B, g.mu, g.delta, g.gamma, g.kappa, cfg->n_channels, d_wtx, d_wrx, d_h, d_pow, nullptr);
        if (g.N == 256) e = wofdm_interf_launch_n256(jobs, g.P, g.B, g.mu, g.delta, g.gamma, g.kappa, cfg->n_channels, d_wtx, d_wrx, d_h, d_pow, nullptr);
        if (g.N == 512) e = wofdm_interf_launch_n512(jobs, g.P, g.B, g.mu, g.delta, g.gamma, g.kappa, cfg->n_channels, d_wtx, d_wrx, d_h, d_pow, nullptr);
        if (g.N == 1024) e = wofdm_interf_launch_n1024(jobs, g.P, g.B, g.mu, g.delta, g.gamma, g.kappa, cfg->n_channels, d_wtx, d_wrx, d_h, d_pow, nullptr);
        if (e != hipSuccess || hipDeviceSynchronize() != hipSuccess ||
            hipMemcpy(power, d_pow, (size_t)jobs * g.N * 4, hipMemcpyDeviceToHost) != hipSuccess) {
            rc = fail(WOFDM_E_HIP, "interference kernel or copy-back failed: %s", hipGetErrorString(hipGetLastError()));
            break;
        }
    } while (0);
    if (d_wtx) (void)hipFree(d_wtx);
    if (d_wrx) (void)hipFree(d_wrx);
    if (d_h) (void)hipFree(d_h);
    if (d_pow) (void)hipFree(d_pow);
    return rc;
}

int wofdm_tx_psd(const wofdm_cfg *cfg, int device, const float *w_tx, const float *X, int no_symbols,
                 int overlap, float *psd)
{
    if (!cfg || !w_tx || !X || !psd) return fail(WOFDM_E_INVALID, "NULL argument");
    const int N = cfg->n_fft;
    if (N != 64 && N != 128 && N != 256)
        return fail(WOFDM_E_UNSUPPORTED, "wofdm_tx_psd is built for n_fft in {64,128,256} (transform length 8 n_fft)");
    const int P = N + cfg->cp + cfg->cs;
    if (cfg->cp < 0 || cfg->cs < 0 || cfg->cp > N || cfg->cs > N || overlap < 0 || 2 * overlap > P || no_symbols < 1)
        return fail(WOFDM_E_INVALID, "bad cp / cs / overlap / no_symbols");
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev)
        return fail(WOFDM_E_HIP, "device %d not available (%d visible); there is no CPU fallback", device, ndev);
    HIP_TRY(hipSetDevice(device));
    const int FL = 8 * N, len = overlap + no_symbols * (P - overlap);
    float *d_w = nullptr, *d_psd = nullptr;
    float2 *d_X = nullptr, *d_x = nullptr;
    int rc = WOFDM_OK;
    do {
        if (hipMalloc(&d_w, (size_t)P * 4) != hipSuccess || hipMalloc(&d_X, (size_t)no_symbols * N * 8) != hipSuccess ||
            hipMalloc(&d_x, (size_t)len * 8) != hipSuccess || hipMalloc(&d_psd, (size_t)FL * 4) != hipSuccess) {
            rc = fail(WOFDM_E_NOMEM, "device allocation failed"); break;
        }
        if (hipMemcpy(d_w, w_tx, (size_t)P * 4, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(d_X, X, (size_t)no_symbols * N * 8, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemset(d_x, 0, (size_t)len * 8) != hipSuccess) {
            rc = fail(WOFDM_E_HIP, "upload failed"); break;
        }
        // (no frame kernel of this process beside these kernels: the gate of launch() is held until they have finished)
        std::lock_guard<std::mutex> gate(g_gate_mu);
        (void)hipDeviceSynchronize();
        hipError_t e = hipErrorInvalidValue;
        if (N == 64) e = wofdm_psd_launch_n64(P, cfg->cp, cfg->cs, overlap, no_symbols, d_w, d_X, d_x, len, d_psd, nullptr);
        if (N == 128) e = wofdm_psd_launch_n128(P, cfg->cp, cfg->cs, overlap, no_symbols, d_w, d_X, d_x, len, d_psd, nullptr);
        if (N == 256) e = wofdm_psd_launch_n256(P, cfg->cp, cfg->cs, overlap, no_symbols, d_w, d_X, d_x, len, d_psd, nullptr);
        if (e != hipSuccess || hipDeviceSynchronize() != hipSuccess ||
            hipMemcpy(psd, d_psd, (size_t)FL * 4, hipMemcpyDeviceToHost) != hipSuccess) {
            rc = fail(WOFDM_E_HIP, "PSD kernels or copy-back failed: %s", hipGetErrorString(hipGetLastError()));
            break;
        }
    } while (0);
    if (d_w) (void)hipFree(d_w);
    if (d_X) (void)hipFree(d_X);
    if (d_x) (void)hipFree(d_x);
    if (d_psd) (void)hipFree(d_psd);
    return rc;
}

int wofdm_philox_kat(int device, const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    if (!ctr || !key || !out) return fail(WOFDM_E_INVALID, "NULL argument");
    HIP_TRY(hipSetDevice(device));
    uint32_t hbuf[6] = {ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1]};
    uint32_t *d = nullptr;
    HIP_TRY(hipMalloc(&d, 10 * sizeof(uint32_t)));
    int rc = WOFDM_OK;
    if (hipMemcpy(d, hbuf, sizeof hbuf, hipMemcpyHostToDevice) != hipSuccess ||
        wofdm_philox_kat_launch(d, d + 6, nullptr) != hipSuccess ||
        hipMemcpy(out, d + 6, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess)
        rc = fail(WOFDM_E_HIP, "philox KAT failed: %s", hipGetErrorString(hipGetLastError()));
    (void)hipFree(d);
    return rc;
}

}  // extern "C"
