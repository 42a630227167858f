// The one-workgroup-per-transform inverse sweep of the map-free PCPS search (pcps_fused.h) and its launch.
// Its own translation unit: at 256 registers per lane the kernel must not let MachineLICM hoist the literal twiddles
// and the loop body's addresses out of the persistent loop (they would be spilled): built with -disable-machine-licm.
// Reference: sydr/dsp/acquisition.py:57-70 (ifft(fft(x) * codeFFT), |.|) and :98-100 (the first maximum).
#include "engine_internal.h"
#include "pcps_codelets.h"

#include <vector>

namespace {
#include "pcps_fused.h"
#include "pcps_fused10k.h"
static_assert(fused25k::kRecordsPerTransform == SDR_PCPS_FUSED_RECORDS, "records per transform");
static_assert(sizeof(fused10k::UnitRecord) == SDR_PCPS_FUSED10K_RECORD_BYTES, "unit record");
}  // namespace

// How the fused sweep cuts a search of n_prn x nbins transforms: whole transforms for the bins that fill whole rounds of
// its 256 workgroups, single rounds for the rest (returns the records per PRN the sweep leaves).
static int fused_plan(int n_prn, int nbins, int* bins_whole, int pieces) {
    const int rounds = (n_prn * nbins) / 256;
    int bw = rounds > 0 ? (256 * rounds) / n_prn : 0;
    if (bw > nbins) bw = nbins;
    // (more than a round's worth left over -- few PRNs, many bins -- : cutting buys nothing, whole transforms throughout)
    if (n_prn * (nbins - bw) > 256) bw = nbins;
    *bins_whole = bw;
    return bw + pieces * (nbins - bw);
}
// units a left-over transform is cut into: its five rounds at N = 25 000; rounds {0, 1}, {2, 3}, {4} at N = 50 000
static int fused_pieces(int terms) { return terms == 2 ? 3 : 5; }

// (terms: 1 at N = 25 000; 2 at N = 50 000, where a unit is one parity of a (PRN, bin) transform and the plan sees
// 2 nbins virtual bins)
int sdr_pcps_fused_records_per_prn(int n_prn, int nbins, int terms) {
    int bw;
    return fused_plan(n_prn, terms * nbins, &bw, fused_pieces(terms)) * SDR_PCPS_FUSED_RECORDS;
}

// C: [n_prn][N] code spectra at N = 25 000; [n_prn][2][N] at N = 50 000 -- the spectrum and its image with the odd half's
// twiddle folded in (pcps.hip code_parity_kernel).
int sdr_pcps_fused_sweep(sdr_engine* e, const void* F, const void* spec_off, const void* C, const void* tw, int n_prn, int nbins, int N,
                         void* partials) {
    if (N != fused25k::N && N != 2 * fused25k::N) return sdr_fail(SDR_ERR_UNSUPPORTED, "fused PCPS sweep: N = %d", N);
    const int terms = N / fused25k::N;
    const int vbins = terms * nbins;
    // (shared spectra: see make_work_list.  Measured, ms of kernels per 32-PRN x 41-bin call with blocks of 4 / 8 / 16 / 32 bins:
    // N = 25 000 0.2165 / 0.2116 / 0.2118 / 0.2144, N = 50 000 0.4834 / 0.4790 / 0.4705 / 0.4665 -- gpurun_out/r05_block_bins.txt)
    const int block_bins = spec_off ? (terms == 2 ? 32 : 16) : 4;
    if (e->pcps_work_prn != n_prn || e->pcps_work_bins != vbins || e->pcps_work_block != block_bins) {
        std::vector<fused25k::WorkItem> order;
        int bins_whole;
        fused_plan(n_prn, vbins, &bins_whole, fused_pieces(terms));
        fused25k::make_work_list(n_prn, vbins, bins_whole, order, e->pcps_work_first, fused_pieces(terms), block_bins);
        if (int rc = sdr_devbuf_reserve(e, &e->pcps_work, order.size() * sizeof(fused25k::WorkItem))) return rc;
        // (pageable source, tiny: the copy is complete when the stream has been waited for)
        SDR_HIP(hipMemcpyAsync(e->pcps_work.ptr, order.data(), order.size() * sizeof(fused25k::WorkItem), hipMemcpyHostToDevice, e->stream));
        SDR_HIP(hipStreamSynchronize(e->stream));
        e->pcps_work_prn = n_prn;
        e->pcps_work_bins = vbins;
        e->pcps_work_block = block_bins;
    }
    fused25k::Args a = {};
    a.spec = (const double2*)F;
    a.spec_off = (const long long*)spec_off;
    a.code_spec = (const double2*)C;
    a.tw = (const double2*)tw;
    a.work = (const fused25k::WorkItem*)e->pcps_work.ptr;
    for (int x = 0; x < 9; ++x) a.xcd_first[x] = e->pcps_work_first[x];
    a.scale = 1.0 / (double)N;
    a.partials = (Best*)partials;
    // the two per-PRN bound arrays (this launch's, the next one's): both zero when made, each launch zeroes the other's
    if (e->pcps_theta_prn != n_prn) {
        if (int rc = sdr_devbuf_reserve(e, &e->pcps_theta, 2 * (size_t)n_prn * sizeof(unsigned long long))) return rc;
        SDR_HIP(hipMemsetAsync(e->pcps_theta.ptr, 0, 2 * (size_t)n_prn * sizeof(unsigned long long), e->stream));
        e->pcps_theta_prn = n_prn;
        e->pcps_theta_flip = 0;
    }
    a.theta = (unsigned long long*)e->pcps_theta.ptr + (size_t)e->pcps_theta_flip * n_prn;
    a.theta_next = (unsigned long long*)e->pcps_theta.ptr + (size_t)(e->pcps_theta_flip ^ 1) * n_prn;
    a.n_prn = n_prn;
    a.nbins = nbins;
    e->pcps_theta_flip ^= 1;
    auto* kernel = terms == 2 ? fused25k::ifft_max_kernel<2> : fused25k::ifft_max_kernel<1>;
    (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fused25k::kLdsBytes);
    ProfScope ps(e, "pcps_inv_fft");
    hipLaunchKernelGGL(kernel, dim3(8 * fused25k::kSlotsPerXcd), dim3(fused25k::kThreads), fused25k::kLdsBytes, e->stream, a);
    SDR_HIP(hipGetLastError());
    return SDR_OK;
}

int sdr_pcps_fused_second(sdr_engine* e, const void* F, const void* spec_off, const void* C, const void* tw, int n_prn, int N, int spc, const void* recs,
                          int per_prn, void* tops, void* dev_bin, void* dev_code, void* seconds, void* res_bin, void* res_code,
                          void* res_ratio) {
    if (N != fused25k::N && N != 2 * fused25k::N) return sdr_fail(SDR_ERR_UNSUPPORTED, "fused PCPS sweep: N = %d", N);
    const int terms = N / fused25k::N;
    fused25k::SecondArgs s = {};
    s.a.spec = (const double2*)F;
    s.a.spec_off = (const long long*)spec_off;
    s.a.code_spec = (const double2*)C;
    s.a.tw = (const double2*)tw;
    s.a.scale = 1.0 / (double)N;
    s.a.partials = (Best*)seconds;
    s.recs = (const Best*)recs;
    s.per_prn = per_prn;
    s.n_prn = n_prn;
    s.spc = spc;
    s.tops = (Best*)tops;
    s.out_bin = (long long*)dev_bin;
    s.out_code = (long long*)dev_code;
    // one ticket per PRN, zero between launches (the workgroup that draws a PRN's last ticket sets it back)
    if (e->pcps_tickets_n < n_prn) {
        if (int rc = sdr_devbuf_reserve(e, &e->pcps_tickets, (size_t)n_prn * sizeof(unsigned))) return rc;
        SDR_HIP(hipMemsetAsync(e->pcps_tickets.ptr, 0, (size_t)n_prn * sizeof(unsigned), e->stream));
        e->pcps_tickets_n = n_prn;
    }
    s.tickets = (unsigned*)e->pcps_tickets.ptr;
    s.res_bin = (long long*)res_bin;
    s.res_code = (long long*)res_code;
    s.res_ratio = (double*)res_ratio;
    if (e->pcps_done && res_bin == e->pcps_res_direct) {      // (the results go straight to the caller's page-locked block)
        s.done = e->pcps_done;
        s.done_seq = e->pcps_done_seq;
        e->pcps_done_used = true;
    }
    auto* kernel = terms == 2 ? fused25k::ifft_second_kernel<2> : fused25k::ifft_second_kernel<1>;
    (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fused25k::kLdsBytes);
    ProfScope ps(e, "pcps_inv_fft");
    // (five workgroups per PRN, a round each, at N = 25 000; six -- two parities x rounds {0, 1}, {2, 3}, {4} -- at N = 50 000)
    hipLaunchKernelGGL(kernel, dim3((terms == 2 ? 6 : 5) * ((n_prn + 7) / 8 * 8)), dim3(fused25k::kThreads), fused25k::kLdsBytes, e->stream, s);
    SDR_HIP(hipGetLastError());
    return SDR_OK;
}

int sdr_pcps_fused10k_search(sdr_engine* e, const void* F_all, const void* spec_off, long long blk_stride, const void* C, const void* tw, int n_prn,
                             int nbins, int noncoh, int N, int spc, void* records, void* out_bin, void* out_code, void* out_ratio) {
    if (N != fused10k::N) return sdr_fail(SDR_ERR_UNSUPPORTED, "fused 10 MHz PCPS search: N = %d", N);
    fused10k::Args a = {};
    a.spec = (const double2*)F_all;
    a.spec_off = (const long long*)spec_off;
    a.blk_stride = blk_stride;
    a.code_spec = (const double2*)C;
    a.tw = (const double2*)tw;
    a.n_prn = n_prn, a.nbins = nbins, a.noncoh = noncoh, a.spc = spc;
    a.scale = 1.0 / (double)N;
    a.records = (fused10k::UnitRecord*)records;
    (void)hipFuncSetAttribute((const void*)fused10k::search_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fused10k::kLdsBytes);
    {
        ProfScope ps(e, "pcps_inv_fft");
        hipLaunchKernelGGL(fused10k::search_kernel, dim3(256), dim3(fused10k::kThreads), fused10k::kLdsBytes, e->stream, a);
    }
    {
        ProfScope ps(e, "pcps_peak");
        unsigned* done = nullptr;
        if (e->pcps_done && out_bin == e->pcps_res_direct) {     // (the results go straight to the caller's page-locked block)
            done = e->pcps_done;
            e->pcps_done_used = true;
        }
        hipLaunchKernelGGL(fused10k::peaks_kernel, dim3(n_prn), dim3(64), 0, e->stream, a.records, nbins, (long long*)out_bin,
                           (long long*)out_code, (double*)out_ratio, done, e->pcps_done_seq);
    }
    SDR_HIP(hipGetLastError());
    return SDR_OK;
}

#ifdef SDR_FUSED_STAMPS
extern "C" __attribute__((visibility("default"))) int sdr_debug_fused_stamps(unsigned long long* out, int reset) {
    if (out) SDR_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(fused25k::g_fused_stamps), sizeof(unsigned long long) * 256 * 8));
    if (reset) {
        static unsigned long long zeros[256 * 8];
        SDR_HIP(hipMemcpyToSymbol(HIP_SYMBOL(fused25k::g_fused_stamps), zeros, sizeof(zeros)));
    }
    return SDR_OK;
}
#endif
