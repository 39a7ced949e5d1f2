// C ABI of the engine (include/rrrmc_hip.h): context, graph marshalling, sampling-call orchestration.
// Product path only: there is no CPU fallback — without a HIP device every compute entry point fails
// with RRRMC_ERR_HIP and a message.
#include "../../include/rrrmc_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <map>
#include <mutex>
#include <utility>
#include <vector>

#include "sk_kernels.hpp"
#include "sk_block_kernel.hpp"
#include "sk_hblock_kernel.hpp"
#include "rrr_kernels.hpp"
#include "quant_wave_kernel.hpp"
#include "sparse_wave_kernel.hpp"
#include "sparse_kernels.hpp"
#include "bign_kernels.hpp"
#include "obs_kernels.hpp"
#include "layout_kernels.hpp"
#include "spf_kernels.hpp"
#include "spf_team_api.hpp"
#include "spf_fast_kernels.hpp"
#include "dbl_kernels.hpp"
#include "cont_kernels.hpp"
#include "cont_wave_kernel.hpp"

using namespace rrrmc;

namespace {

thread_local std::string g_create_error;

constexpr int64_t kMaxSlotsPerBatch = 1 << 22;    // plan buffers: 2 sets x 2 x 16 MiB; one sweep launch per batch
constexpr int64_t kMaxChunksPerBatch = 1 << 16;
constexpr int kLdsLimit = 160 * 1024;
constexpr int kMaxChunk = kMaxChunkSlots;          // longest chunk (slots per pipeline step) the sweep kernel is given
#ifdef RRRMC_STAMPS
unsigned long long* g_stamps = nullptr;
#endif

}  // namespace

struct rrrmc_ctx {
    int32_t model = 0;
    int64_t N = 0, K = 0, R = 0;
    int64_t G = 0, Rpad = 0;
    int device = 0;
    uint32_t replica0 = 0;
    hipStream_t stream = nullptr;
    uint64_t seed = 0;
    bool seeded = false;
    uint64_t it_done = 0;
    bool graph_set = false, spins_set = false;

    // disorder
    int32_t* d_A = nullptr;
    int8_t* d_J = nullptr;
    uint16_t* d_table = nullptr;
    int TS = 0;
    // state
    uint32_t* d_spins = nullptr;   // [G][N]
    int32_t* d_E = nullptr;        // [Rpad]
    int64_t* d_acc = nullptr;      // [Rpad]
    // plan buffers
    ChunkDesc* d_chunks = nullptr;
    ChunkDesc* d_chunks_alt = nullptr;   // a second copy of the table: the planner fills in per-chunk fields, and the planner of a call runs beside the previous call's sweeps
    size_t chunks_cap = 0;
    ChunkDesc* h_chunks = nullptr; // pinned staging buffer of the chunk list
    size_t h_chunks_cap = 0;
    // the chunk list is call-relative (ChunkDesc::g0), i.e. a function of (iters, step, C) only: the list on the device is kept
    // across calls of the same shape, and the pinned buffer is rewritten only after its last upload has completed (ev_upload)
    int64_t chunks_iters = -1, chunks_step = -1;
    int chunks_C = -1;
    size_t chunks_n = 0;
    bool upload_pending = false;
    typedef ChunkBatch BatchDesc;
    std::vector<BatchDesc> chunk_batches;
    uint32_t* d_slots[2] = {nullptr, nullptr};   // double-buffered: the planner of batch b+1 overlaps the sweep of batch b
    uint32_t* d_vecs[2] = {nullptr, nullptr};
    uint32_t* d_nbrs[2] = {nullptr, nullptr};    // big mode: the neighbour row of every slot (K words), written by plan_big_kernel
    uint32_t* d_masks[2] = {nullptr, nullptr};   // big mode, few groups: acceptance masks of every (slot, group) of a batch (big_mask_kernel)
    bool big_masks = false;
    bool big_cm = false;                         // compact masks: one word per (slot, workgroup)
    int big_lgr = 5;                             // big_apply_kernel: 2^lgr replicas per workgroup (their spins fit LDS)
    uint32_t* d_bigimg = nullptr;                // [G * S][ceil(N / S)] the workgroups' LDS images between big_apply_kernel and big_merge_kernel
    int64_t batch_chunks_max = kMaxChunksPerBatch;
    int64_t batch_first_chunks = 0;              // > 0: the first batch of a call holds at most this many chunks
    int64_t batch_slots_max = kMaxSlotsPerBatch; // slots per batch (plan buffers; the mask buffers when big_masks)
    hipStream_t plan_stream = nullptr;
    hipEvent_t ev_upload = nullptr;
    std::vector<hipEvent_t> ev_plan;     // plan of batch b finished
    // the two sets of plan buffers across calls: the set a call's first batch takes alternates, and a set is written again only behind the
    // sweep that read it last — so the planner of a call's first batch runs beside the previous call's last sweep
    hipEvent_t ev_set_free[2] = {nullptr, nullptr};
    bool set_used[2] = {false, false};
    int plan_parity = 0;
    int call_parity = 0;
    // staging of the caller-layout transfers: BitVector chunks (rrrmc_set_spins / rrrmc_get_spins / rrrmc_snapshot_get) and the
    // replica-major energy samples (rrrmc_fetch_results); kept across calls, grown on demand
    unsigned long long* d_io = nullptr;
    size_t io_cap = 0;                   // bytes
    // results of the last sampling call
    int32_t* d_Es = nullptr;
    size_t Es_cap = 0;
    int64_t nsamp = 0;
    bool results_valid = false;
    // sweep geometry
    int C = 0;
    size_t lds_bytes = 0, plan_lds_bytes = 0;
    // timing
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    std::vector<hipEvent_t> ev_sweep;   // pairs
    int sweep_launches = 0;
    bool timing_valid = false;
    // accumulation mode (rrrmc_timing_accumulate): every sweep launch of every sampling call gets its own event pair from a growing
    // pool, so that a caller can queue many async calls and read the summed kernel time afterwards without a sync per call
    bool acc_mode = false;
    std::vector<hipEvent_t> ev_pool;
    size_t ev_pool_used = 0;
    size_t last_ev_base = 0;            // first event of the last call's sweeps (in ev_pool when last_ev_pool, else ev_sweep)
    bool last_ev_pool = false;

    // ---- colour-parallel sweeps (any sparse ctx; the only sampler when lds_mode is false) ----
    bool lds_mode = true;               // the LDS-resident random-site kernel is available (the state fits the 160 KiB LDS)
    bool wide = false;                  // the neighbour table stays in HBM/L2 (longer chunks; needed for 8192 < N)
    bool big_bufs_ready = false;        // ensure_big_buffers has run
    bool big_mode = false;              // N does not fit LDS: random-site standardMC through plan_big_kernel / big_sweep_kernel (spins in HBM/L2)
    int sweep_mode = 0;                 // sweep_kernel<K, MODE>: 0 LDS table / byte offsets, 1 HBM table / byte offsets, 2 HBM table / word indices
    int ncolors = 0;
    std::vector<int32_t> color_count;   // sites per colour
    std::vector<uint32_t*> d_color_list; // device lists: one record per site of the colour (site, neighbour row with coupling signs)
    uint32_t* d_U = nullptr;            // [Rpad] unsatisfied-bond counters of the bit-sliced energy kernel
    uint64_t sweeps_done = 0;
    std::vector<int32_t> h_A;           // host copy of the neighbour table (colouring check)
    std::vector<uint8_t> h_Jsign;       // host copy of the coupling signs (1 = J < 0), +-J models: colour records
    bool colored_call = false;          // the last sampling call was a colored-sweep call
    bool color_count_acc = false;       // colour sweeps also count every replica's accepted moves (rrrmc_colored_count_accepted)
    // ---- RRRMC_MODEL_SK_NORMAL ----
    double* sk_J = nullptr;        // [N][N]
    double* sk_J4 = nullptr;       // [N][ldJ] 4 J (exact): what update_cache! adds (SK.jl:256-262); RRRMC_MODEL_SK_BINARY: the +-4.0 matrix, built from the bit rows at the first standardMC call
    bool skb_J4_valid = false;
    double* sk_blkJw = nullptr;    // sk_block_kernel: coupling sub-matrices of a segment's blocks
    uint32_t* sk_blkSites = nullptr;
    size_t sk_blk_cap = 0;         // blocks the two buffers hold
    double* sk_hl = nullptr;       // [G8 * 8][N] sk_hblock_kernel's lfields_last scratch (valid inside a launch only)
    double* sk_lf = nullptr;       // [G8][N][8]
    double* sk_lfl = nullptr;
    int32_t* sk_move_last = nullptr;
    uint8_t* sk_spins = nullptr;   // [G8][N]
    double* sk_E = nullptr;        // [G8 * 8]
    double* sk_Es = nullptr;
    size_t sk_Es_cap = 0;
    int64_t G8 = 0;
    // ---- RRRMC_MODEL_SK_BINARY (shares the SK buffers above except the fields / couplings) ----
    uint32_t* skb_J = nullptr;     // [N][NW]
    int32_t* skb_lf = nullptr;
    int32_t* skb_lfl = nullptr;
    int skb_NW = 0;
    // ---- RRRMC_MODEL_QUANT_RRG ----
    int64_t qM = 0, qNk = 0, qW = 0;      // Trotter slices, spins per slice, 32-bit words per replica
    uint32_t* q_Jb = nullptr;             // GraphQuant over binary GraphSK slices (GraphQSKT): [Nk][q_Wk] words of J's rows, else null
    int64_t q_Wk = 0;
    bool q_sk = false;
    bool q_skn = false;                   // GraphQuant over GraphSKNormal slices (GraphQSKNormalT): couplings in sk_J [Nk][Nk], slice caches below
    double* q_slf = nullptr;              // [R][2][M][Nk]
    int32_t* q_smv = nullptr;             // [R][M]
    uint8_t* q_scur = nullptr;            // [R][M]
    bool q_spf = false;                   // GraphQuant over sparse Float64 slices (GraphQEAT, src/QAliases.jl:50-83): table in d_A, couplings and slice caches below
    double* q_Jf = nullptr;               // [Nk][K]
    double* q_flf = nullptr;              // [R][M][Nk]
    double* q_fundo = nullptr;            // [R][M][K+1]
    int32_t* q_fml = nullptr;             // [R][M]
    uint32_t* q_spins = nullptr;
    uint8_t* q_cls = nullptr;
    uint16_t* q_sv = nullptr;
    uint16_t* q_spos = nullptr;
    int32_t* q_st = nullptr;
    double* q_T = nullptr;
    double* q_z = nullptr;
    double* q_accrate = nullptr;
    int64_t* q_stats = nullptr;
    double last_fourK = 0.0, last_beta = 0.0;
    bool last_call_rrr = false;
    int stats_stride = 2;
    // ---- rrrMC / bklMC on RRRMC_MODEL_SPARSE_PM1 (allocated on first use) ----
    uint32_t* rp_spins = nullptr;
    uint8_t* rp_cls = nullptr;
    uint16_t* rp_sv = nullptr;
    uint16_t* rp_spos = nullptr;
    // ---- wtmMC on RRRMC_MODEL_SPARSE_PM1 (allocated on first use) ----
    double* wt_t = nullptr;        // [R][N] heap keys
    uint16_t* wt_id = nullptr;
    uint16_t* wt_pos = nullptr;
    double* wt_time = nullptr;     // [R] final global time of the last call
    uint32_t wtm_calls = 0;        // wtmMC calls since the last rrrmc_seed (part of the WTM stream address)
    bool last_call_wtm = false;
    // ---- extremal_opt on RRRMC_MODEL_SPARSE_PM1 (allocated on first use; shares the rrrMC class arrays) ----
    uint32_t* eo_cmin = nullptr;   // [R][2*nch] configuration of minimum energy, BitVector word order
    double* eo_ftau = nullptr;     // [N]
    bool last_call_eo = false;
    // ---- rrrMC(SingleGraph) on RRRMC_MODEL_SK_NORMAL: DeltaECacheCont + DynamicSampler state (allocated on first use) ----
    double* rs_buf = nullptr;      // lfA, lfB, v, ps, dEs, st_dE, st_p, z_out
    uint32_t* rs_spins = nullptr;
    int32_t* rs_status = nullptr;

    // ---- RRRMC_MODEL_SPARSE_F64 (GraphRRGNormal / GraphEANormal): shares sk_lf / sk_lfl / sk_move_last / sk_E / sk_Es / d_acc ----
    double* pf_J = nullptr;        // [N][K]
    unsigned long long* pf_spins = nullptr;  // [W][N]: bit l = spin of replica 64 w + l
    double* pf_undo = nullptr;     // [W][K+1][64]: live part of lfields_last (see spf_kernels.hpp)
    int32_t* pf_sites = nullptr;   // site stream of one launch
    uint32_t* pf_plan = nullptr;   // spf_team_kernel: the attempts of one launch (spf_team_kernel.hpp), allocated at its first use
    int32_t* pf_status = nullptr;  // spf_team_kernel: set by a launch whose wait ran into its limit (checked by rrrmc_sync)
    int64_t pfW = 0;
    bool pf_multi_edge = false;    // some row of A repeats a neighbour (GraphEANormal with L = 2): the wave build of the continuous samplers is not used
    bool pf_lf_live = false;       // sk_lf holds the local fields of the current configuration (false after the continuous samplers)
    // ---- RRRMC_MODEL_SPARSE_DISCRETIZED (Graph{RRG,EA}NormalDiscretized): spins in q_spins / qW (BitVector word order),
    //      energies in sk_E / sk_Es, statistics in q_stats ----
    int8_t* db_dJ = nullptr;       // [N][K] discretised couplings (levels)
    double* db_rJ = nullptr;       // [N][K] residuals
    uint8_t* db_cls = nullptr;
    uint16_t* db_sv = nullptr;
    uint16_t* db_spos = nullptr;
    double* db_lf = nullptr;       // [R][N]
    double* db_undo = nullptr;     // [R][K+1]
    int db_L = 0, db_ea_form = 0;
    long long db_lev_mul = 1;          // level units -> Float64: (units * mul) / div (DFloat64 levels: rrrmc_set_level_scale)
    double db_lev_div = 1.0;
    bool db_cache_valid = false;
    bool q_lds_attr = false;            // hipFuncSetAttribute done for the LDS-resident rrrMC kernel
    bool q_cache_valid = false;         // GraphQuant: the DeltaECache arrays describe the last rrrMC call
    // ---- level table of the integer-level kernels (rrr_sparse / wtm / eo / lev_standard): the +-J table for RRRMC_MODEL_SPARSE_PM1,
    // allΔE(X) of the given levels for RRRMC_MODEL_SPARSE_LEVELS (whose spins live in q_spins / qW, BitVector word order)
    LevTable lv{};
    long long lv_mul = 1;
    double lv_div = 1.0;
    int64_t eo_W = 0;                  // words per replica of eo_cmin
    int db_dElist[kDLmax] = {0};
    // ---- continuous-energy rrrMC / bklMC / wtmMC on RRRMC_MODEL_SPARSE_F64 (allocated on first use) ----
    uint32_t* cs_spins = nullptr;  // [R][W] replica-contiguous words
    double* cs_buf = nullptr;      // lf, dEs (heap keys), v, ps, undo
    uint16_t* cs_u16 = nullptr;    // heap id / position
    // ---- snapshots / observables ----
    uint8_t* snap = nullptr;       // [nslots][snap_bytes]: copies of the model's native spin buffer
    int32_t snap_slots = 0;
    std::vector<uint8_t> snap_valid;
    void** d_pairs = nullptr;      // [2][pairs_cap] device pointer tables of the overlap kernels
    size_t pairs_cap = 0;
    int32_t* d_ovl = nullptr;      // [pairs_cap][Rpad]
    int32_t* d_qobs = nullptr;     // GraphQuant: e0[R], Eslice[R][M], ovs_raw[R][M/2]

    // ---- fast standardMC on RRRMC_MODEL_SPARSE_F64 (spf_fast_kernels.hpp; allocated on first use) ----
    std::vector<double> h_Jf;           // host copy of the couplings (threshold tables per beta)
    bool pff_ready = false, pff_beta_valid = false;
    double pff_beta = 0.0;
    int pff_C = 0;
    size_t pff_lds = 0, pff_plan_lds = 0;
    uint16_t* pff_table = nullptr;
    uint32_t* pff_thr_hi = nullptr; uint32_t* pff_thr_lo = nullptr; uint32_t* pff_flags = nullptr;
    double* pff_absJ = nullptr;         // the energy phase's tables of partial sums (etab)
    uint32_t* pff_bond_off = nullptr;
    int pff_nblk = 0;
    // ---- resumed standardMC calls (rrrmc_set_resume): a hooked run of a Float64 model is ONE chain (src/RRRMC.jl:95-118) ----
    bool resume = false;                // the next standardMC calls continue from the tracked energy and the live cache
    bool std_cache_live = false;        // the model's cache (fields, undo record) and tracked energy describe the current configuration
    int32_t* db_mlast = nullptr;        // [R] move_last of the residual cache (RRRMC_MODEL_SPARSE_DISCRETIZED), kept across resumed calls

    // ---- resumed rrrMC / bklMC / wtmMC / extremal_opt calls (rrrmc_set_resume): the RUN whose chain state is live on the device ----
    // (the cache arrays of the sampler's kernels plus the two scalar slabs of rrr_kernels.hpp: SmpState)
    int smp_kind = 0;                   // 0 none, 1 rrrMC, 2 bklMC, 3 wtmMC, 4 extremal_opt
    double smp_par[4] = {0, 0, 0, 0};   // the run's parameters: beta, staged_thr (wtmMC: step), staged_thr_fact, fourK
    int64_t smp_step = 0;               // bklMC: step (part of the loop's state: nextstep advances by it)
    uint64_t smp_g0 = 0;                // it_done when the run began (bklMC numbers its moves from there)
    int64_t smp_it = 0;                 // iterations of the run so far (wtmMC: samples)
    uint32_t smp_call = 0;              // wtmMC: the WTM stream's call number of the run
    int smp_build = 0;                  // which kernel build holds the state, where the builds' layouts differ (cont_wave_kernel = 1)
    std::vector<double> smp_ftau;       // extremal_opt: the run's rank table
    double* smp_f = nullptr;            // [R][kSmpF]
    long long* smp_i = nullptr;         // [R][kSmpI]
    double* cs_top = nullptr;           // cont_wave_kernel: [R][N2 / 16] LDS levels of the sampler's tree between calls

    // ---- multi-device context (rrrmc_ctx_create_multi): no device state of its own; one child context per device (own stream), replica
    //      shards of whole 32-replica groups in global-id order; every entry point forwards to the children ----
    // ---- debug mode (rrrmc_set_debug_checks): the reference's latent consistency checks (src/graphs/RRG.jl:229-231, SK.jl:268-273),
    //      run on the device after every standardMC call: tracked energy == energy(X, C), cached fields == recomputed fields ----
    bool debug_checks = false;
    bool multi_poisoned = false;        // multi-device parent: a forwarded call failed on some children only
    int32_t* dbg_flag = nullptr;        // [2]: number of replicas that failed the last check, one of them
    int32_t* dbg_Ei = nullptr;          // [Rpad] tracked energies (integer models)
    double* dbg_lf = nullptr; double* dbg_lfl = nullptr; double* dbg_E = nullptr; int32_t* dbg_ml = nullptr;    // recomputed SK cache
    std::vector<rrrmc_ctx*> kids;
    std::vector<int64_t> kid_r0;        // first replica of each child, relative to this context's first

    std::string err;
};

namespace {

int32_t fail(rrrmc_ctx* ctx, int32_t code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf; else g_create_error = buf;
    return code;
}

#define HIP_TRY(ctx, expr)                                                                              \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return fail(ctx, RRRMC_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

template <typename T> void free_dev(T*& p) { if (p) { (void)hipFree(p); p = nullptr; } }

// template instantiation tables: runtime K (or sites-per-thread) -> kernel pointer
#define RRRMC_DISPATCH_UPTO7(k, F)                                                                                     \
    switch (k) { case 1: return F<1>; case 2: return F<2>; case 3: return F<3>; case 4: return F<4>; case 5: return F<5>;   \
                 case 6: return F<6>; case 7: return F<7>; default: return nullptr; }
#define RRRMC_DISPATCH_UPTO8(k, F)                                                                                     \
    switch (k) { case 1: return F<1>; case 2: return F<2>; case 3: return F<3>; case 4: return F<4>; case 5: return F<5>;   \
                 case 6: return F<6>; case 7: return F<7>; case 8: return F<8>; default: return nullptr; }

typedef void (*plan_fn)(ChunkDesc*, uint32_t*, uint32_t*, const int32_t*, int, int, uint32_t, uint32_t, uint64_t, int);
// rows of 64 slots per consumer batch of sweep_kernel<K>
int sweep_rows_for_K(int K) { switch (K) { case 1: return sweep_rows<1>(); case 2: return sweep_rows<2>(); case 3: return sweep_rows<3>(); case 4: return sweep_rows<4>(); case 5: return sweep_rows<5>(); case 6: return sweep_rows<6>(); default: return sweep_rows<7>(); } }
plan_fn plan_for_K(int K) { RRRMC_DISPATCH_UPTO7(K, plan_kernel) }

typedef void (*plan_big_fn)(ChunkDesc*, uint32_t*, uint32_t*, uint32_t*, const int32_t*, const int8_t*, int, uint32_t, uint32_t, uint64_t, int);
plan_big_fn plan_big_for_K(int K) { RRRMC_DISPATCH_UPTO7(K, plan_big_kernel) }
typedef void (*big_sweep_fn)(BigSweepParams);
big_sweep_fn big_sweep_for_K(int K) { RRRMC_DISPATCH_UPTO7(K, big_sweep_kernel) }
typedef void (*big_mask_fn)(BigMaskParams);
big_mask_fn big_mask_for_K(int K) { RRRMC_DISPATCH_UPTO7(K, big_mask_kernel) }
typedef void (*big_apply_fn)(BigSweepParams, const uint32_t*, uint32_t*, uint32_t, int);
big_apply_fn big_apply_for_K(int K) { RRRMC_DISPATCH_UPTO7(K, big_apply_kernel) }
big_apply_fn big_applyc_for_K(int K) { RRRMC_DISPATCH_UPTO7(K, big_applyc_kernel) }

typedef void (*sweep_fn)(SweepParams);
sweep_fn sweep_for_K(int K, int mode)
{
    // mode 0: table in LDS, byte offsets; 1: table in HBM/L2, byte offsets (N <= 8192); 2: table in HBM/L2, word indices;
    // 3: table in HBM/L2, one word per site, word indices with the coupling sign in bit 15 (N <= 32767)
    switch (mode * 8 + K) {
        case 1: return sweep_kernel<1, 0>; case 2: return sweep_kernel<2, 0>; case 3: return sweep_kernel<3, 0>; case 4: return sweep_kernel<4, 0>;
        case 5: return sweep_kernel<5, 0>; case 6: return sweep_kernel<6, 0>; case 7: return sweep_kernel<7, 0>;
        case 9: return sweep_kernel<1, 1>; case 10: return sweep_kernel<2, 1>; case 11: return sweep_kernel<3, 1>; case 12: return sweep_kernel<4, 1>;
        case 13: return sweep_kernel<5, 1>; case 14: return sweep_kernel<6, 1>; case 15: return sweep_kernel<7, 1>;
        case 17: return sweep_kernel<1, 2>; case 18: return sweep_kernel<2, 2>; case 19: return sweep_kernel<3, 2>; case 20: return sweep_kernel<4, 2>;
        case 21: return sweep_kernel<5, 2>; case 22: return sweep_kernel<6, 2>; case 23: return sweep_kernel<7, 2>;
        case 25: return sweep_kernel<1, 3>; case 26: return sweep_kernel<2, 3>; case 27: return sweep_kernel<3, 3>; case 28: return sweep_kernel<4, 3>;
        case 29: return sweep_kernel<5, 3>; case 30: return sweep_kernel<6, 3>; case 31: return sweep_kernel<7, 3>;
        default: return nullptr;
    }
}

// wide = the neighbour table is not staged in LDS (8192 < N: offsets are word indices, see SweepParams::table)
size_t sweep_lds_bytes(int64_t N, int K, int TS, int C, bool wide = false, bool single = false)
{
    const int NT = (K + 1) / 2, NW = NT + (K + 2) / 2, NQ = (NW + 3) / 4;
    const size_t words = (size_t)(((single ? 1 : 2) * N + 128 + 3) & ~3ll) + (size_t)3 * NQ * 4 * C + (size_t)8 * (C + 64) + (size_t)(4 + 2 * kSwLeftMax * (1 + 2 * NT));
    return words * 4 + (wide ? (size_t)0 : (size_t)N * TS * 2);
}

size_t plan_lds_bytes(int64_t N, int K, int C)
{
    // max(s_bin [N+1], s_cnt + s_start + s_cur [3 (C+2)]) u32 (two phases share it); s_site, s_lvl, s_item [C], s_nb [C*K], s_pred [C*(K+1)] u16
    const size_t share = std::max((size_t)N + 1, 3 * ((size_t)C + 2));
    return 4 * share + 2 * ((size_t)3 * C + (size_t)C * K + (size_t)C * (K + 1)) + 16;
}

// Launch shape of the thread-per-replica kernels (rrr_*, wtm, eo, cont, dbl): these chains are latency bound and each lane walks
// its own replica's arrays, so a full 64-lane wavefront issues 64 scattered requests per load.  With few replicas one replica per
// workgroup spreads them over the CUs (measured at 128 replicas: 1.6x on GraphQuant, 2.5x on the discretised DoubleGraph);
// with many, up to 64 per workgroup.  RRRMC_RRR_TPB overrides (timing experiments).
inline unsigned rrr_tpb(int64_t R)
{
    if (const char* e = std::getenv("RRRMC_RRR_TPB")) { const int v = std::atoi(e); if (v >= 1 && v <= kRrrThreads) return (unsigned)v; }
    // about one wavefront per CU (256 on MI355X): measured best at 128 / 4096 / 16384 replicas, within 5 % at 1024
    unsigned t = 1u;
    while (t < (unsigned)kRrrThreads && (int64_t)t * 256 < R) t *= 2u;
    return t;
}
inline unsigned rrr_blocks(int64_t R) { const unsigned t = rrr_tpb(R); return (unsigned)((R + t - 1) / t); }

// models whose device spins are R x W 32-bit words in BitVector order (q_spins, qW)
// The dynamic-LDS bound is an attribute of the KERNEL, not of a context: contexts of different sizes share it, so it is only ever
// raised (a later, smaller context must not lower it under an earlier one).
// hipFuncSetAttribute acts on the CURRENT device's copy of the kernel, so the bound is tracked per (device, kernel): a process
// with one context per device raises it on each of them.  Callers have made the context's device current.
hipError_t raise_lds_attr(const void* fn, size_t bytes)
{
    static std::mutex mu;
    static std::map<std::pair<int, const void*>, size_t> cur;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(mu);
    size_t& have = cur[std::make_pair(dev, fn)];
    if (bytes <= have) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) have = bytes;
    return e;
}

inline bool chunk_layout(const rrrmc_ctx* ctx) { return ctx->model == RRRMC_MODEL_QUANT_RRG || ctx->model == RRRMC_MODEL_SPARSE_DISCRETIZED || ctx->model == RRRMC_MODEL_SPARSE_LEVELS; }
inline bool sparse_int_model(const rrrmc_ctx* ctx) { return ctx->model == RRRMC_MODEL_SPARSE_PM1 || ctx->model == RRRMC_MODEL_SPARSE_LEVELS; }
inline double lv_to_f64(const rrrmc_ctx* ctx, long long units) { return (double)(units * ctx->lv_mul) / ctx->lv_div; }

int32_t ensure_state(rrrmc_ctx* ctx, bool need_spins)
{
    if (!ctx) return RRRMC_ERR_INVALID_ARG;
    if (!ctx->graph_set) return fail(ctx, RRRMC_ERR_STATE, "rrrmc_set_graph has not been called");
    if (need_spins && !ctx->spins_set)
        return fail(ctx, RRRMC_ERR_STATE, "no configuration: call rrrmc_init_spins_random or rrrmc_set_spins first");
    return RRRMC_OK;
}

// the caller-layout staging buffer (ctx->d_io) holds at least `bytes`
int32_t ensure_io(rrrmc_ctx* ctx, size_t bytes)
{
    if (bytes <= ctx->io_cap) return RRRMC_OK;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    free_dev(ctx->d_io);
    ctx->io_cap = 0;
    if (hipMalloc(&ctx->d_io, bytes) != hipSuccess) { ctx->d_io = nullptr; return fail(ctx, RRRMC_ERR_NOMEM, "cannot allocate %zu bytes of transfer staging", bytes); }
    ctx->io_cap = bytes;
    return RRRMC_OK;
}

typedef void (*ebs_fn)(const uint32_t*, const int32_t*, const int8_t*, int, uint32_t*);
ebs_fn energy_bs_for_K(int K) { RRRMC_DISPATCH_UPTO7(K, energy_bs_kernel) }
#ifndef RRRMC_COLOR_GPT
#define RRRMC_COLOR_GPT 1
#endif
// replica groups per thread of the plain colour-sweep kernel (the counting build: 1).  Measured on GraphEA(64, 3), 512 replicas:
// 1: 2.60, 2: 2.65, 4: 2.51, 8: 2.37 x 10^12 attempts/s — sharing the neighbour row between groups buys nothing
constexpr int kColorGPT = RRRMC_COLOR_GPT;
typedef void (*csweep_fn)(ColorSweepParams);
csweep_fn csweep_for_K(int K, bool count)
{
    if (count) {
        switch (K) { case 1: return colored_sweep_kernel<1, true>; case 2: return colored_sweep_kernel<2, true>; case 3: return colored_sweep_kernel<3, true>;
                     case 4: return colored_sweep_kernel<4, true>; case 5: return colored_sweep_kernel<5, true>; case 6: return colored_sweep_kernel<6, true>;
                     case 7: return colored_sweep_kernel<7, true>; default: return nullptr; }
    }
    switch (K) { case 1: return colored_sweep_kernel<1, false, kColorGPT>; case 2: return colored_sweep_kernel<2, false, kColorGPT>; case 3: return colored_sweep_kernel<3, false, kColorGPT>;
                 case 4: return colored_sweep_kernel<4, false, kColorGPT>; case 5: return colored_sweep_kernel<5, false, kColorGPT>; case 6: return colored_sweep_kernel<6, false, kColorGPT>;
                 case 7: return colored_sweep_kernel<7, false, kColorGPT>; default: return nullptr; }
}

// bit-sliced energy of every replica into d_E (and, optionally, one row of the sample buffer)
int32_t run_energy_bs(rrrmc_ctx* ctx, int32_t* es_row)
{
    const int64_t per = 256 * kEnergySitesPerThread;
    const dim3 grid((unsigned)((ctx->N + per - 1) / per), (unsigned)ctx->G);
    hipLaunchKernelGGL(energy_bs_for_K((int)ctx->K), grid, dim3(256), 0, ctx->stream, ctx->d_spins, ctx->d_A, ctx->d_J, (int)ctx->N, ctx->d_U);
    HIP_TRY(ctx, hipGetLastError());
    hipLaunchKernelGGL(energy_bs_finish_kernel, dim3((unsigned)((ctx->Rpad + 255) / 256)), dim3(256), 0, ctx->stream, ctx->d_U, ctx->d_E, es_row,
                       (int)ctx->Rpad, (int)(ctx->N * ctx->K / 2));
    HIP_TRY(ctx, hipGetLastError());
    return RRRMC_OK;
}

int32_t run_energy(rrrmc_ctx* ctx, uint8_t* d_nun)
{
    if (!d_nun) return run_energy_bs(ctx, nullptr);
    hipLaunchKernelGGL(energy_kernel, dim3((unsigned)ctx->G), dim3(256), 0, ctx->stream, ctx->d_spins, ctx->d_A, ctx->d_J,
                       (int)ctx->N, (int)ctx->K, ctx->d_E, d_nun);
    HIP_TRY(ctx, hipGetLastError());
    return RRRMC_OK;
}


// ---- debug mode: consistency checks on the device ----------------------------------------------------------------------------
__global__ void dbg_compare_i32_kernel(const int32_t* a, const int32_t* b, int n, int32_t* flag)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n && a[r] != b[r]) { atomicAdd(&flag[0], 1); flag[1] = r; }
}
// |a - b| <= tol * scale element-wise; replica of element e = (e / inner) * rb + e % rb (the [G][N][8] field layout; inner = N * 8)
__global__ void dbg_compare_f64_kernel(const double* a, const double* b, long long n, double tol, long long inner, int rb, int32_t* flag)
{
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const double d = a[e] - b[e];
    if (!(d <= tol && d >= -tol)) { atomicAdd(&flag[0], 1); flag[1] = (int32_t)((e / inner) * rb + e % rb); }
}
__global__ void dbg_inject_i32_kernel(int32_t* a) { a[0] += 2; }          // fault injection for the test of the failing branch
__global__ void dbg_inject_f64_kernel(double* a) { a[0] += 1.0; }
inline bool dbg_inject() { const char* e = std::getenv("RRRMC_DEBUG_INJECT"); return e && e[0] == '1'; }
// after a standardMC call of the +-J model: d_E (tracked) against energy(X, C) recomputed from the spins
int32_t debug_check_pm1(rrrmc_ctx* ctx)
{
    // (the flag is zeroed when it is allocated and by rrrmc_sync after reading it, never here: a mismatch found after an earlier queued call
    //  must survive until the sync that reports it)
    if (!ctx->dbg_flag) { HIP_TRY(ctx, hipMalloc(&ctx->dbg_flag, sizeof(int32_t) * 2)); HIP_TRY(ctx, hipMemsetAsync(ctx->dbg_flag, 0, sizeof(int32_t) * 2, ctx->stream)); }
    if (!ctx->dbg_Ei) HIP_TRY(ctx, hipMalloc(&ctx->dbg_Ei, sizeof(int32_t) * ctx->Rpad));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->dbg_Ei, ctx->d_E, sizeof(int32_t) * ctx->Rpad, hipMemcpyDeviceToDevice, ctx->stream));
    if (dbg_inject()) hipLaunchKernelGGL(dbg_inject_i32_kernel, dim3(1), dim3(1), 0, ctx->stream, ctx->dbg_Ei);       // (RRRMC_DEBUG_INJECT=1: tests only)
    const int32_t rc = run_energy_bs(ctx, nullptr);          // d_E = energy(X, C): equal to the tracked values, or the flag says so
    if (rc) return rc;
    hipLaunchKernelGGL(dbg_compare_i32_kernel, dim3((unsigned)((ctx->R + 255) / 256)), dim3(256), 0, ctx->stream, ctx->dbg_Ei, ctx->d_E, (int)ctx->R, ctx->dbg_flag);
    HIP_TRY(ctx, hipGetLastError());
    return RRRMC_OK;
}
// ---- resumed calls of the reduced-rejection samplers ---------------------------------------------------------------------------------
// anything that changes the configuration, the disorder, the streams' position or the caches behind the samplers' back ends the run
inline void smp_drop(rrrmc_ctx* ctx) { if (ctx) ctx->smp_kind = 0; }
// Start of a sampler call of `kind` with the given parameters: *S describes it to the kernel.  The call CONTINUES the live run when the
// context is in resume mode and the run is of the same sampler with the same parameters (the weights of the cache depend on them);
// otherwise it starts a run as a reference call does: energy(X, C), a fresh cache (src/RRRMC.jl:175-178).
int32_t smp_begin(rrrmc_ctx* ctx, int kind, double p0, double p1, double p2, double p3, int64_t step, const double* ftau, SmpState* S)
{
    if (!ctx->smp_f) {
        HIP_TRY(ctx, hipMalloc(&ctx->smp_f, sizeof(double) * (size_t)ctx->R * kSmpF));
        HIP_TRY(ctx, hipMalloc(&ctx->smp_i, sizeof(long long) * (size_t)ctx->R * kSmpI));
        HIP_TRY(ctx, hipMemsetAsync(ctx->smp_f, 0, sizeof(double) * (size_t)ctx->R * kSmpF, ctx->stream));
        HIP_TRY(ctx, hipMemsetAsync(ctx->smp_i, 0, sizeof(long long) * (size_t)ctx->R * kSmpI, ctx->stream));
    }
    bool cont = ctx->resume && ctx->smp_kind == kind && ctx->smp_par[0] == p0 && ctx->smp_par[1] == p1 && ctx->smp_par[2] == p2 &&
                ctx->smp_par[3] == p3 && (kind != 2 || ctx->smp_step == step);
    if (cont && kind == 4)
        cont = ctx->smp_ftau.size() == (size_t)ctx->N && std::memcmp(ctx->smp_ftau.data(), ftau, sizeof(double) * (size_t)ctx->N) == 0;
    ctx->smp_kind = 0;                  // live again once the call has been queued (smp_commit)
    if (!cont) {
        ctx->smp_par[0] = p0; ctx->smp_par[1] = p1; ctx->smp_par[2] = p2; ctx->smp_par[3] = p3;
        ctx->smp_step = step; ctx->smp_g0 = ctx->it_done; ctx->smp_it = 0; ctx->smp_call = ctx->wtm_calls; ctx->smp_build = 0;
        if (kind == 4) ctx->smp_ftau.assign(ftau, ftau + ctx->N);
    }
    S->sf = ctx->smp_f; S->si = ctx->smp_i; S->resume = cont ? 1 : 0;
    S->it0 = ctx->smp_it;
    S->samp0 = step > 0 ? step - ctx->smp_it % step : 1;
    return RRRMC_OK;
}
// samples a call of `n` more iterations takes, the run being at iteration smp_it (multiples of `step` of the RUN's iteration count)
inline int64_t smp_nsamp(const rrrmc_ctx* ctx, int64_t n, int64_t step) { return (ctx->smp_it + n) / step - ctx->smp_it / step; }
inline void smp_commit(rrrmc_ctx* ctx, int kind, int64_t n) { ctx->smp_kind = kind; ctx->smp_it += n; }

#include "host_sk.hpp"
#include "host_spf.hpp"
#include "host_dbl.hpp"
#include "host_rrr.hpp"
#include "host_lev.hpp"
#include "host_sweep.hpp"
#include "host_spf_fast.hpp"
int32_t quant_mc_async(rrrmc_ctx* ctx, bool standard, double beta, double fourK, int64_t iters, int64_t step, double staged_thr, double staged_thr_fact);

}  // namespace

namespace {
// ---- multi-device contexts -------------------------------------------------------------------------------------------------
inline bool is_multi(const rrrmc_ctx* c) { return c && !c->kids.empty(); }
template <typename T> inline T* at_row(T* p, int64_t n) { return p ? p + n : nullptr; }
inline int64_t nch_of(const rrrmc_ctx* c) { return (c->N + 63) / 64; }
// f(child, first replica of the child, replicas of the child) on every child: one after the other for calls that only enqueue work on
// the child's stream (the devices then run concurrently), one host thread per device for calls that wait or copy.  The first failure is
// reported with the device it came from.
template <typename F> int32_t multi_each(rrrmc_ctx* ctx, F&& f, bool threads)
{
    if (ctx->multi_poisoned)
        return fail(ctx, RRRMC_ERR_STATE, "an earlier call failed on some devices of this multi-device context only: its shards are at different positions; destroy it");
    const size_t n = ctx->kids.size();
    std::vector<int32_t> rc(n, RRRMC_OK);
    if (threads && n > 1) {
        std::vector<std::thread> th;
        th.reserve(n);
        for (size_t d = 0; d < n; ++d) th.emplace_back([&, d] { rc[d] = f(ctx->kids[d], ctx->kid_r0[d], ctx->kids[d]->R); });
        for (std::thread& t : th) t.join();
    } else {
        for (size_t d = 0; d < n; ++d) { rc[d] = f(ctx->kids[d], ctx->kid_r0[d], ctx->kids[d]->R); if (rc[d]) { if (d > 0) ctx->multi_poisoned = true; break; } }
    }
    for (size_t d = 0; d < n; ++d)
        if (rc[d]) {
            ctx->err = "device " + std::to_string(ctx->kids[d]->device) + " (replicas from " + std::to_string(ctx->kid_r0[d]) + "): " + ctx->kids[d]->err;
            // the other shards may have run (or queued) the call: unless every shard refused it the same way (an argument error is
            // detected by all of them before anything is queued), the shards are no longer at one stream position
            if (threads && n > 1) {
                bool same = true;
                for (size_t e = 0; e < n; ++e) same = same && rc[e] == rc[d];
                if (!same) ctx->multi_poisoned = true;
            }
            return rc[d];
        }
    return RRRMC_OK;
}
}  // namespace
// forward to the children of a multi-device context; `c`, `r0` (the child's first replica) and `rn` (its replicas) are visible in `expr`
#define RRRMC_MULTI(ctx, threads, expr)                                                                                  \
    do {                                                                                                                 \
        if (is_multi(ctx))                                                                                               \
            return multi_each(ctx, [&](rrrmc_ctx* c, int64_t r0, int64_t rn) -> int32_t { (void)r0; (void)rn; return (expr); }, threads); \
    } while (0)

namespace {
// What a finished sampling call may have left for the host to report: the debug mode's mismatch flag, spf_team_kernel's protocol failure, the
// dynamic sampler's loss of precision.  Called by every entry point that has synchronised the stream and is about to hand results or state
// back (rrrmc_sync, rrrmc_fetch_results*, rrrmc_get_spins, rrrmc_tracked_energy*): none of them returns the output of a void call as good.
int32_t post_sync_checks(rrrmc_ctx* ctx)
{
    if (ctx->debug_checks && ctx->dbg_flag) {
        int32_t fl[2] = {0, 0};
        HIP_TRY(ctx, hipMemcpy(fl, ctx->dbg_flag, sizeof fl, hipMemcpyDeviceToHost));
        HIP_TRY(ctx, hipMemset(ctx->dbg_flag, 0, sizeof fl));
        if (fl[0])
            return fail(ctx, RRRMC_ERR_STATE, "debug check failed for %d value(s), e.g. replica %d: the tracked energy / cached fields differ from "
                                              "energy(X, C) recomputed from the configuration (src/graphs/RRG.jl:229-231, SK.jl:268-273)", fl[0], fl[1]);
    }
    if (ctx->model == RRRMC_MODEL_SPARSE_F64 && ctx->pf_status && !ctx->last_call_rrr) {
        int32_t st = 0;
        HIP_TRY(ctx, hipMemcpy(&st, ctx->pf_status, sizeof st, hipMemcpyDeviceToHost));
        if (st) {
            HIP_TRY(ctx, hipMemset(ctx->pf_status, 0, sizeof st));
            ctx->results_valid = false; ctx->std_cache_live = false; ctx->pf_lf_live = false;
            return fail(ctx, RRRMC_ERR_STATE, "spf_team_kernel: a wait on the retired prefix ran into its limit (protocol failure); the call's results are void. "
                                              "RRRMC_SPF_TEAM=0 selects the single-wavefront kernel");
        }
    }
    if (ctx->last_call_rrr && (ctx->model == RRRMC_MODEL_SK_NORMAL || ctx->model == RRRMC_MODEL_SPARSE_F64) && ctx->rs_status) {
        std::vector<int32_t> stt((size_t)ctx->R);
        HIP_TRY(ctx, hipMemcpy(stt.data(), ctx->rs_status, sizeof(int32_t) * stt.size(), hipMemcpyDeviceToHost));
        for (int64_t r = 0; r < ctx->R; ++r)
            if (stt[r]) return fail(ctx, RRRMC_ERR_STATE, "replica %lld: Unrecoverable loss of precision detected in the dynamic sampler", (long long)r);   // DynamicSamplers.jl:147
    }
    return RRRMC_OK;
}
}  // namespace

extern "C" {





int32_t rrrmc_version(void) { return 100; }

const char* rrrmc_last_error(const rrrmc_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int32_t rrrmc_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int32_t rrrmc_device_copy_bandwidth(int32_t device, int64_t nbytes, int32_t reps, double* gbps_out)
{
    if (!gbps_out || nbytes < 1 || reps < 1) return fail(nullptr, RRRMC_ERR_INVALID_ARG, "gbps_out is NULL or nbytes / reps < 1");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(nullptr, RRRMC_ERR_HIP, "no HIP device is visible");
    if (device < 0 || device >= ndev) return fail(nullptr, RRRMC_ERR_INVALID_ARG, "device %d out of range (0..%d)", device, ndev - 1);
    void *a = nullptr, *b = nullptr;
    hipStream_t st = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    float ms = 0.f;
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = hipMalloc(&a, (size_t)nbytes);
    if (e == hipSuccess) e = hipMalloc(&b, (size_t)nbytes);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    if (e == hipSuccess) e = hipMemsetAsync(a, 1, (size_t)nbytes, st);
    if (e == hipSuccess) e = hipMemcpyAsync(b, a, (size_t)nbytes, hipMemcpyDeviceToDevice, st);          // warm-up
    if (e == hipSuccess) e = hipEventRecord(e0, st);
    for (int i = 0; i < reps && e == hipSuccess; ++i) e = hipMemcpyAsync(b, a, (size_t)nbytes, hipMemcpyDeviceToDevice, st);
    if (e == hipSuccess) e = hipEventRecord(e1, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (st) (void)hipStreamDestroy(st);
    if (a) (void)hipFree(a);
    if (b) (void)hipFree(b);
    if (e != hipSuccess) return fail(nullptr, e == hipErrorOutOfMemory ? RRRMC_ERR_NOMEM : RRRMC_ERR_HIP, "device copy timing failed: %s", hipGetErrorString(e));
    *gbps_out = 2.0 * (double)nbytes * (double)reps / ((double)ms * 1e-3) / 1e9;          // bytes read + bytes written
    return RRRMC_OK;
}

int32_t rrrmc_host_alloc(int64_t nbytes, void** out)
{
    if (!out || nbytes < 0) return RRRMC_ERR_INVALID_ARG;
    *out = nullptr;
    if (nbytes == 0) return RRRMC_OK;
    void* p = nullptr;
    if (hipHostMalloc(&p, (size_t)nbytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return RRRMC_ERR_NOMEM; }
    *out = p;
    return RRRMC_OK;
}

int32_t rrrmc_host_free(void* p)
{
    if (!p) return RRRMC_OK;
    return hipHostFree(p) == hipSuccess ? RRRMC_OK : RRRMC_ERR_HIP;
}

int32_t rrrmc_ctx_create(rrrmc_ctx** out, int32_t model, int64_t N, int64_t K, int64_t R, int32_t device, uint32_t replica0)
{
    if (!out) return fail(nullptr, RRRMC_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    if (model == RRRMC_MODEL_QUANT_RRG) return fail(nullptr, RRRMC_ERR_INVALID_ARG, "RRRMC_MODEL_QUANT_RRG contexts are created with rrrmc_ctx_create_quant");
    if (model != RRRMC_MODEL_SPARSE_PM1 && model != RRRMC_MODEL_SK_NORMAL && model != RRRMC_MODEL_SK_BINARY && model != RRRMC_MODEL_SPARSE_F64 &&
        model != RRRMC_MODEL_SPARSE_DISCRETIZED && model != RRRMC_MODEL_SPARSE_LEVELS)
        return fail(nullptr, RRRMC_ERR_UNSUPPORTED, "unknown model kind %d", model);
    if (model == RRRMC_MODEL_SPARSE_LEVELS) return lev_ctx_create(out, N, K, R, device, replica0);
    if (model == RRRMC_MODEL_SPARSE_F64) return spf_ctx_create(out, N, K, R, device, replica0);
    if (model == RRRMC_MODEL_SPARSE_DISCRETIZED) return dbl_ctx_create(out, N, K, R, device, replica0);
    if (model == RRRMC_MODEL_SK_NORMAL || model == RRRMC_MODEL_SK_BINARY) return sk_ctx_create(out, model, N, R, device, replica0);
    if (N < 1 || K < 1 || R < 1) return fail(nullptr, RRRMC_ERR_INVALID_ARG, "N, K, R must be >= 1 (given N=%lld K=%lld R=%lld)", (long long)N, (long long)K, (long long)R);
    if (K > kMaxK) return fail(nullptr, RRRMC_ERR_UNSUPPORTED, "K=%lld: the sparse +-J kernels cover K <= %d", (long long)K, kMaxK);
    if (N > (int64_t)INT32_MAX / 8) return fail(nullptr, RRRMC_ERR_UNSUPPORTED, "N=%lld is too large", (long long)N);
    if (replica0 % 32) return fail(nullptr, RRRMC_ERR_INVALID_ARG, "replica0 must be a multiple of 32 (given %u)", replica0);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, RRRMC_ERR_HIP, "no HIP device is visible: this library has no CPU path");
    if (device < 0 || device >= ndev) return fail(nullptr, RRRMC_ERR_INVALID_ARG, "device %d out of range (0..%d)", device, ndev - 1);

    rrrmc_ctx* ctx = new (std::nothrow) rrrmc_ctx();
    if (!ctx) return fail(nullptr, RRRMC_ERR_NOMEM, "out of host memory");
    ctx->model = model; ctx->N = N; ctx->K = K; ctx->R = R;
    ctx->G = (R + 31) / 32; ctx->Rpad = ctx->G * 32;
    ctx->device = device; ctx->replica0 = replica0;
    ctx->TS = K <= 4 ? 4 : 8;

    // chunk length: the longest multiple of 64 that fits the 160 KiB LDS next to the state (a step of the sweep kernel has a fixed
    // cost of about a microsecond, so long chunks pay: 2.7 ns per slot at C = 1472 against 3.2 at 832), capped by kMaxChunk.
    // WIDE build (word indices instead of byte offsets, neighbour table in HBM/L2 instead of LDS): needed when byte offsets into
    // the 2N-word LDS spin array no longer fit 16 bits (N > 8192); chosen as well when dropping the LDS copy of the table buys a
    // clearly longer chunk (at equal chunk length it is 1.6 % slower than the normal build).
    auto max_chunk = [&](bool wide, bool single = false) {
#ifdef RRRMC_CHUNK_TASKS
        int c = RRRMC_CHUNK_TASKS * kWave;
#else
        int c = kMaxChunk;
#endif
        while (c >= kWave && ((single ? N > 32767 : 2 * N + 128 > 65535) || (!wide && N > 8192) ||
                              sweep_lds_bytes(N, (int)K, ctx->TS, c, wide, single) > (size_t)kLdsLimit)) c -= kWave;
        return c;
    };
    const int Cn = max_chunk(false), Cw = max_chunk(true), Cs = N > 8192 ? max_chunk(true, true) : 0;
    ctx->wide = Cw >= Cn + Cn / 8;
    if (const char* fw = std::getenv("RRRMC_FORCE_WIDE")) ctx->wide = ctx->wide || fw[0] == '1';       // timing experiments
    ctx->sweep_mode = !ctx->wide ? 0 : (N <= 8192 ? 1 : 2);
    int C = ctx->wide ? Cw : Cn;
    // one word per site (the consumer pays three more instructions per neighbour): only where it buys clearly longer chunks
    bool single = N > 8192 && N <= 32767 && Cs >= 4 * kWave && Cs >= C + C / 2;
    if (const char* fs = std::getenv("RRRMC_FORCE_SINGLE")) {      // tests / timing experiments: "1" forces it at any N <= 32767, "0" forbids it
        if (fs[0] == '1' && N <= 32767) single = true;
        if (fs[0] == '0') single = false;
    }
    if (single) { ctx->wide = true; ctx->sweep_mode = 3; C = max_chunk(true, true); }
    // tiny graphs: the dependency depth of a chunk grows like C (K+1) / N, and both the planner's relaxation rounds and the
    // consumer's sequential levels follow it — long chunks stop paying
    const int64_t cap = std::max<int64_t>(4 * kWave, (8 * N) / kWave * kWave);
    if (C > cap) C = (int)cap;
    // otherwise only the colour-parallel sweeps are available (HBM/L2-resident spins)
    ctx->lds_mode = C >= 4 * kWave && plan_lds_bytes(N, (int)K, C) <= (size_t)kLdsLimit;
    if (const char* fb = std::getenv("RRRMC_FORCE_BIG")) { if (fb[0] == '1') ctx->lds_mode = false; }      // tests: the big-N path at small N
    ctx->big_mode = !ctx->lds_mode && N <= ((int64_t)1 << kBigSiteBits);
    if (!ctx->lds_mode) C = ctx->big_mode ? kBigChunk : kWave;
    ctx->C = C;
    ctx->lds_bytes = ctx->lds_mode ? sweep_lds_bytes(N, (int)K, ctx->TS, C, ctx->wide, ctx->sweep_mode == 3) : 0;
    ctx->plan_lds_bytes = ctx->lds_mode ? plan_lds_bytes(N, (int)K, C) : (ctx->big_mode ? plan_big_lds_bytes((int)K) : 0);
    ctx->batch_slots_max = kMaxSlotsPerBatch;
    if (ctx->big_mode) {
        // few replica groups: the acceptance masks of a batch are made by the whole device first (big_mask_kernel), the per-group
        // workgroups only apply them.  The mask buffers (2 x at most 4 GiB) bound the batch.
        ctx->big_lgr = big_lds_lgr(N);
        ctx->big_cm = big_masks_compact((int)K, ctx->big_lgr);
        // four mask words per (slot, group), or one per (slot, workgroup) where a workgroup's mask bits fit a word
        const int64_t per_slot = (int64_t)ctx->G * (ctx->big_cm ? 4 * (32 >> ctx->big_lgr) : 16);
        int64_t fit = ((int64_t)1 << 32) / per_slot / kBigChunk * kBigChunk;
        ctx->big_masks = ctx->G <= 512 && fit >= 16 * kBigChunk;
        if (const char* e = std::getenv("RRRMC_BIG_NO_MASKS")) { if (e[0] == '1') ctx->big_masks = false; }
        if (ctx->big_masks) {
            // a launch of big_apply_kernel takes at most kBigApplyChunks chunks: no reason to hold masks for more
            if (fit > (int64_t)kBigApplyChunks * kBigChunk) fit = (int64_t)kBigApplyChunks * kBigChunk;
            if (fit < ctx->batch_slots_max) ctx->batch_slots_max = fit;
        }
        if (ctx->big_masks) { ctx->batch_chunks_max = kBigApplyChunks; ctx->batch_first_chunks = 64; }
    }

#define CREATE_TRY(expr)                                                                                         \
    do {                                                                                                         \
        hipError_t e_ = (expr);                                                                                  \
        if (e_ != hipSuccess) {                                                                                  \
            int32_t rc_ = fail(nullptr, RRRMC_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));           \
            rrrmc_ctx_destroy(ctx);                                                                              \
            return rc_;                                                                                          \
        }                                                                                                        \
    } while (0)
    CREATE_TRY(hipSetDevice(device));
    CREATE_TRY(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    CREATE_TRY(hipEventCreate(&ctx->ev_begin));
    CREATE_TRY(hipEventCreate(&ctx->ev_end));
    CREATE_TRY(hipMalloc(&ctx->d_A, sizeof(int32_t) * N * K));
    CREATE_TRY(hipMalloc(&ctx->d_J, sizeof(int8_t) * N * K));
    CREATE_TRY(hipMalloc(&ctx->d_table, sizeof(uint16_t) * (ctx->lds_mode ? N * ctx->TS : 8)));
    CREATE_TRY(hipMalloc(&ctx->d_U, sizeof(uint32_t) * ctx->Rpad));
    CREATE_TRY(hipMemset(ctx->d_U, 0, sizeof(uint32_t) * ctx->Rpad));
    CREATE_TRY(hipMalloc(&ctx->d_spins, sizeof(uint32_t) * ctx->G * N));
    CREATE_TRY(hipMalloc(&ctx->d_E, sizeof(int32_t) * ctx->Rpad));
    CREATE_TRY(hipMalloc(&ctx->d_acc, sizeof(int64_t) * ctx->Rpad));
    for (int i = 0; i < 2; ++i) {
        CREATE_TRY(hipMalloc(&ctx->d_slots[i], sizeof(uint32_t) * kMaxSlotsPerBatch));
        CREATE_TRY(hipMalloc(&ctx->d_vecs[i], sizeof(uint32_t) * kMaxSlotsPerBatch));
    }
    // (the record / mask / image buffers of the big-N random-site path — up to 2 x 128 MiB + 2 x 2 GiB — are allocated by its first
    //  call, ensure_big_buffers: a context that only runs colour sweeps never pays for them)
    CREATE_TRY(hipStreamCreateWithFlags(&ctx->plan_stream, hipStreamNonBlocking));
    CREATE_TRY(hipEventCreateWithFlags(&ctx->ev_upload, hipEventDisableTiming));
    CREATE_TRY(hipMemset(ctx->d_spins, 0, sizeof(uint32_t) * ctx->G * N));
    CREATE_TRY(hipMemset(ctx->d_E, 0, sizeof(int32_t) * ctx->Rpad));
    CREATE_TRY(hipMemset(ctx->d_acc, 0, sizeof(int64_t) * ctx->Rpad));
    if (ctx->lds_mode) {
        sweep_fn fn = sweep_for_K((int)K, ctx->sweep_mode);
        CREATE_TRY(raise_lds_attr(reinterpret_cast<const void*>(fn), ctx->lds_bytes));
        CREATE_TRY(raise_lds_attr(reinterpret_cast<const void*>(plan_for_K((int)K)), ctx->plan_lds_bytes));
    }
    if (ctx->big_mode)
        CREATE_TRY(raise_lds_attr(reinterpret_cast<const void*>(plan_big_for_K((int)K)), ctx->plan_lds_bytes));
#undef CREATE_TRY
    *out = ctx;
    return RRRMC_OK;
}

void rrrmc_ctx_destroy(rrrmc_ctx* ctx)
{
    if (!ctx) return;
    if (is_multi(ctx)) {
        for (rrrmc_ctx* c : ctx->kids) rrrmc_ctx_destroy(c);
        delete ctx;
        return;
    }
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    free_dev(ctx->d_A); free_dev(ctx->d_J); free_dev(ctx->d_table); free_dev(ctx->d_spins);
    free_dev(ctx->d_E); free_dev(ctx->d_acc); free_dev(ctx->d_chunks); free_dev(ctx->d_chunks_alt); free_dev(ctx->d_Es);
    free_dev(ctx->d_U); free_dev(ctx->d_io);
    for (uint32_t*& l : ctx->d_color_list) free_dev(l);
    free_dev(ctx->sk_J4); free_dev(ctx->sk_blkJw); free_dev(ctx->sk_blkSites); free_dev(ctx->sk_hl);
    free_dev(ctx->sk_J); free_dev(ctx->sk_lf); free_dev(ctx->sk_lfl); free_dev(ctx->sk_move_last); free_dev(ctx->sk_spins);
    free_dev(ctx->sk_E); free_dev(ctx->sk_Es); free_dev(ctx->skb_J); free_dev(ctx->skb_lf); free_dev(ctx->skb_lfl);
    free_dev(ctx->q_spins); free_dev(ctx->q_cls); free_dev(ctx->q_sv); free_dev(ctx->q_spos); free_dev(ctx->q_st);
    free_dev(ctx->q_T); free_dev(ctx->q_z); free_dev(ctx->q_accrate); free_dev(ctx->q_stats);
    free_dev(ctx->rs_buf); free_dev(ctx->rs_spins); free_dev(ctx->rs_status);
    free_dev(ctx->rp_spins); free_dev(ctx->rp_cls); free_dev(ctx->rp_sv); free_dev(ctx->rp_spos);
    free_dev(ctx->pf_J); free_dev(ctx->pf_spins); free_dev(ctx->pf_undo); free_dev(ctx->pf_sites); free_dev(ctx->pf_plan); free_dev(ctx->pf_status);
    free_dev(ctx->db_dJ); free_dev(ctx->db_rJ); free_dev(ctx->db_cls); free_dev(ctx->db_sv); free_dev(ctx->db_spos); free_dev(ctx->db_lf); free_dev(ctx->db_undo); free_dev(ctx->db_mlast);
    free_dev(ctx->pff_table); free_dev(ctx->pff_absJ); free_dev(ctx->pff_bond_off); free_dev(ctx->pff_thr_hi); free_dev(ctx->pff_thr_lo); free_dev(ctx->pff_flags);
    free_dev(ctx->wt_t); free_dev(ctx->wt_id); free_dev(ctx->wt_pos); free_dev(ctx->wt_time);
    free_dev(ctx->smp_f); free_dev(ctx->smp_i); free_dev(ctx->cs_top);
    free_dev(ctx->q_Jf); free_dev(ctx->q_flf); free_dev(ctx->q_fundo); free_dev(ctx->q_fml);
    free_dev(ctx->eo_cmin); free_dev(ctx->eo_ftau);
    free_dev(ctx->q_Jb); free_dev(ctx->q_slf); free_dev(ctx->q_smv); free_dev(ctx->q_scur);
    free_dev(ctx->cs_spins); free_dev(ctx->cs_buf); free_dev(ctx->cs_u16);
    free_dev(ctx->snap); free_dev(ctx->d_pairs); free_dev(ctx->d_ovl); free_dev(ctx->d_qobs);
    for (int i = 0; i < 2; ++i) { free_dev(ctx->d_slots[i]); free_dev(ctx->d_vecs[i]); free_dev(ctx->d_nbrs[i]); free_dev(ctx->d_masks[i]); }
    free_dev(ctx->d_bigimg);
    free_dev(ctx->dbg_flag); free_dev(ctx->dbg_Ei); free_dev(ctx->dbg_lf); free_dev(ctx->dbg_lfl); free_dev(ctx->dbg_E); free_dev(ctx->dbg_ml);
    if (ctx->plan_stream) { (void)hipStreamSynchronize(ctx->plan_stream); (void)hipStreamDestroy(ctx->plan_stream); }
    if (ctx->ev_upload) (void)hipEventDestroy(ctx->ev_upload);
    for (hipEvent_t e : ctx->ev_plan) (void)hipEventDestroy(e);
    for (int i = 0; i < 2; ++i) if (ctx->ev_set_free[i]) (void)hipEventDestroy(ctx->ev_set_free[i]);
    if (ctx->h_chunks) (void)hipHostFree(ctx->h_chunks);
    for (hipEvent_t e : ctx->ev_pool) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->ev_sweep) (void)hipEventDestroy(e);
    if (ctx->ev_begin) (void)hipEventDestroy(ctx->ev_begin);
    if (ctx->ev_end) (void)hipEventDestroy(ctx->ev_end);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int32_t rrrmc_set_graph(rrrmc_ctx* ctx, const int32_t* A, const int8_t* J)
{
    RRRMC_MULTI(ctx, false, rrrmc_set_graph(c, A, J));
    smp_drop(ctx);
    if (!ctx) return RRRMC_ERR_INVALID_ARG;
    if (ctx->model != RRRMC_MODEL_SPARSE_PM1 && ctx->model != RRRMC_MODEL_QUANT_RRG)
        return fail(ctx, RRRMC_ERR_STATE, "rrrmc_set_graph is for sparse +-J models; use rrrmc_set_couplings_dense");
    if (ctx->model == RRRMC_MODEL_QUANT_RRG && ctx->q_sk)
        return fail(ctx, RRRMC_ERR_STATE, "this GraphQuant has binary GraphSK slices: give their couplings with rrrmc_set_couplings_bits");
    if (!A || !J) return fail(ctx, RRRMC_ERR_INVALID_ARG, "A and J must not be NULL");
    const int64_t N = ctx->model == RRRMC_MODEL_QUANT_RRG ? ctx->qNk : ctx->N, K = ctx->K;
    for (int64_t q = 0; q < N * K; ++q) {
        if (A[q] < 0 || A[q] >= N) return fail(ctx, RRRMC_ERR_INVALID_ARG, "A[%lld] = %d out of range 0..%lld", (long long)q, A[q], (long long)(N - 1));
        if (J[q] != 1 && J[q] != -1)    // GraphRRG ctor: "the given J is incompatible with levels" RRG.jl:130
            return fail(ctx, RRRMC_ERR_INVALID_ARG, "J[%lld] = %d: couplings must be -1 or +1", (long long)q, (int)J[q]);
        if (A[q] == q / K) return fail(ctx, RRRMC_ERR_INVALID_ARG, "self loop at site %lld", (long long)(q / K));
    }
    // every bond must appear from both ends with the same coupling (multi-edges allowed: GraphEA with L = 2)
    std::vector<uint8_t> used((size_t)(N * K), 0);
    for (int64_t x = 0; x < N; ++x)
        for (int64_t k = 0; k < K; ++k) {
            const int64_t y = A[x * K + k];
            bool found = false;
            for (int64_t l = 0; l < K && !found; ++l)
                if (!used[y * K + l] && A[y * K + l] == x && J[y * K + l] == J[x * K + k]) { used[y * K + l] = 1; found = true; }
            if (!found) return fail(ctx, RRRMC_ERR_INVALID_ARG, "bond (%lld,%lld) is not symmetric in (A, J)", (long long)x, (long long)y);
        }
    if (ctx->model == RRRMC_MODEL_QUANT_RRG) {
        // Slice graphs that list a neighbour twice (GraphEA with L = 2: two bonds to the same site, EA.jl:24-43,156): delta_energy sums
        // every entry, neighbors(X1[k], i) is the de-duplicated list uA (EA.jl:158, :292) — GraphEA's rules, announced by the caller with
        // rrrmc_quant_slice_form(ctx, 1).  A GraphRRG keeps repeated entries in uA (RRG.jl:133): not covered, refused.
        for (int64_t x = 0; x < N && !ctx->db_ea_form; ++x)
            for (int64_t k = 0; k < K; ++k)
                for (int64_t l = k + 1; l < K; ++l)
                    if (A[x * K + k] == A[x * K + l])
                        return fail(ctx, RRRMC_ERR_UNSUPPORTED, "site %lld lists neighbour %d twice: only GraphEA slices may (call rrrmc_quant_slice_form(ctx, 1) first)", (long long)x, A[x * K + k]);
        for (int64_t x = 0; x < N && ctx->db_ea_form; ++x)
            for (int64_t k = 1; k < K; ++k)
                if (A[x * K + k] < A[x * K + k - 1])
                    return fail(ctx, RRRMC_ERR_INVALID_ARG, "GraphEA slices: the neighbour rows must be ascending (EA.jl:24-43), violated at site %lld", (long long)x);
        HIP_TRY(ctx, hipSetDevice(ctx->device));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        HIP_TRY(ctx, hipMemcpy(ctx->d_A, A, sizeof(int32_t) * N * K, hipMemcpyHostToDevice));
        HIP_TRY(ctx, hipMemcpy(ctx->d_J, J, sizeof(int8_t) * N * K, hipMemcpyHostToDevice));
        ctx->graph_set = true;
        return RRRMC_OK;
    }
    ctx->lv = LevTable{};
    ctx->lv.L = (int)(K / 2 + 1);                       // allΔE has K/2 + 1 levels for +-J couplings (RRG.jl:262-265, EA.jl:293)
    for (int k = 0; k < ctx->lv.L && k < kSLmax; ++k) ctx->lv.dElist[k] = 2 * (2 * k + (int)(K & 1));
    std::vector<uint16_t> table((size_t)(ctx->lds_mode ? N * ctx->TS : 8), 0);
    if (ctx->lds_mode)
        for (int64_t x = 0; x < N; ++x)
            for (int64_t k = 0; k < K; ++k)
                table[x * ctx->TS + k] = ctx->sweep_mode == 3 ? (uint16_t)(A[x * K + k] | (J[x * K + k] < 0 ? 0x8000 : 0))      // one word per site, sign in bit 15
                                                               : (uint16_t)((ctx->sweep_mode == 2 ? 1 : 4) * (2 * A[x * K + k] + (J[x * K + k] < 0 ? 1 : 0)));   // byte offset (word index in mode 2) in the LDS spin array: word 2y = s_y, word 2y + 1 = ~s_y
    ctx->h_A.assign(A, A + N * K);
    ctx->h_Jsign.resize((size_t)(N * K));
    for (int64_t e = 0; e < N * K; ++e) ctx->h_Jsign[e] = J[e] < 0 ? 1 : 0;
    ctx->ncolors = 0;                                           // a colouring holds records of the graph it was set for: set it again
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(ctx->d_A, A, sizeof(int32_t) * N * K, hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(ctx->d_J, J, sizeof(int8_t) * N * K, hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(ctx->d_table, table.data(), sizeof(uint16_t) * table.size(), hipMemcpyHostToDevice));
    ctx->graph_set = true;
    return RRRMC_OK;
}

int32_t rrrmc_seed(rrrmc_ctx* ctx, uint64_t seed)
{
    RRRMC_MULTI(ctx, false, rrrmc_seed(c, seed));
    smp_drop(ctx);
    if (!ctx) return RRRMC_ERR_INVALID_ARG;
    ctx->seed = seed;
    ctx->seeded = true;
    ctx->it_done = 0;
    ctx->sweeps_done = 0;
    ctx->wtm_calls = 0;
    return RRRMC_OK;
}

// which build of spf_team_kernel the default standardMC of a RRRMC_MODEL_SPARSE_F64 context launches (bench.py names the kernel its traffic
// figures belong to from this, not from a copy of the selection rule)
int32_t rrrmc_spf_team_build(rrrmc_ctx* ctx, int32_t* waves_out, int32_t* width_out, int32_t* slots_out)
{
    if (!ctx) return RRRMC_ERR_INVALID_ARG;
    if (is_multi(ctx)) return rrrmc_spf_team_build(ctx->kids[0], waves_out, width_out, slots_out);
    if (ctx->model != RRRMC_MODEL_SPARSE_F64) return fail(ctx, RRRMC_ERR_STATE, "rrrmc_spf_team_build is for RRRMC_MODEL_SPARSE_F64");
    const bool team = spf_use_team(ctx);
    const SpfTeamBuild tb = team ? spf_team_build(ctx) : SpfTeamBuild{0, 0, 0};
    if (waves_out) *waves_out = tb.lds ? tb.nw : 0;          // 0: the one-wavefront spf_sweep_kernel runs (RRRMC_SPF_TEAM=0, or no build for this K)
    if (width_out) *width_out = tb.lds ? tb.tw : 0;
    if (slots_out) *slots_out = tb.lds ? spf_team_slots((int)ctx->K, tb.nw, tb.tw) : 0;
    return RRRMC_OK;
}
int64_t rrrmc_results_samples(const rrrmc_ctx* ctx) { return !ctx ? -1 : is_multi(ctx) ? ctx->kids[0]->nsamp : ctx->nsamp; }
int64_t rrrmc_iterations_done(const rrrmc_ctx* ctx) { return !ctx ? -1 : is_multi(ctx) ? (int64_t)ctx->kids[0]->it_done : (int64_t)ctx->it_done; }

int32_t rrrmc_init_spins_random(rrrmc_ctx* ctx)
{
    RRRMC_MULTI(ctx, false, rrrmc_init_spins_random(c));
    smp_drop(ctx);
    if (!ctx) return RRRMC_ERR_INVALID_ARG;
    ctx->std_cache_live = false;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (chunk_layout(ctx)) {
        const dim3 grid((unsigned)((ctx->qW + 255) / 256), (unsigned)ctx->R);
        hipLaunchKernelGGL(quant_init_spins_kernel, grid, dim3(256), 0, ctx->stream, ctx->q_spins, (int)ctx->N, (int)ctx->qW,
                           ctx->replica0, (uint32_t)ctx->seed, (uint32_t)(ctx->seed >> 32));
        HIP_TRY(ctx, hipGetLastError());
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        ctx->spins_set = true;
        return RRRMC_OK;
    }
    if (ctx->model == RRRMC_MODEL_SK_NORMAL || ctx->model == RRRMC_MODEL_SK_BINARY) {
        const dim3 grid((unsigned)((ctx->N + 255) / 256), (unsigned)ctx->G8);
        hipLaunchKernelGGL(sk_init_spins_kernel, grid, dim3(256), 0, ctx->stream, ctx->sk_spins, (int)ctx->N, ctx->replica0,
                           (uint32_t)ctx->seed, (uint32_t)(ctx->seed >> 32));
        HIP_TRY(ctx, hipGetLastError());
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        ctx->spins_set = true;
        return RRRMC_OK;
    }
    if (ctx->model == RRRMC_MODEL_SPARSE_F64) {
        const dim3 grid((unsigned)((ctx->N + 255) / 256), (unsigned)ctx->pfW);
        hipLaunchKernelGGL(spf_init_spins_kernel, grid, dim3(256), 0, ctx->stream, ctx->pf_spins, (int)ctx->N, ctx->replica0,
                           (uint32_t)ctx->seed, (uint32_t)(ctx->seed >> 32));
        HIP_TRY(ctx, hipGetLastError());
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        ctx->pf_lf_live = false;
        ctx->spins_set = true;
        return RRRMC_OK;
    }
    const dim3 grid((unsigned)((ctx->N + 255) / 256), (unsigned)ctx->G);
    hipLaunchKernelGGL(init_spins_kernel, grid, dim3(256), 0, ctx->stream, ctx->d_spins, (int)ctx->N, ctx->replica0 / 32,
                       (uint32_t)ctx->seed, (uint32_t)(ctx->seed >> 32));
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->spins_set = true;
    return RRRMC_OK;
}

int32_t rrrmc_set_spins(rrrmc_ctx* ctx, const uint64_t* chunks)
{
    RRRMC_MULTI(ctx, true, rrrmc_set_spins(c, at_row(chunks, r0 * nch_of(ctx))));
    smp_drop(ctx);
    if (ctx) ctx->std_cache_live = false;
    if (!ctx) return RRRMC_ERR_INVALID_ARG;
    if (!chunks) return fail(ctx, RRRMC_ERR_INVALID_ARG, "chunks is NULL");
    const int64_t N = ctx->N, nch = (N + 63) / 64;
    if (N % 64) {   // BitVector invariant: unused bits of the last chunk are zero
        const uint64_t tailmask = ~0ull << (N % 64);
        for (int64_t r = 0; r < ctx->R; ++r)
            if (chunks[r * nch + nch - 1] & tailmask)
                return fail(ctx, RRRMC_ERR_INVALID_ARG, "replica %lld: bits beyond N are set in the last chunk", (long long)r);
    }
    if (chunk_layout(ctx)) {     // chunk c of a replica = words 2c, 2c+1 (little endian): same bytes
        HIP_TRY(ctx, hipSetDevice(ctx->device));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        HIP_TRY(ctx, hipMemcpy(ctx->q_spins, chunks, sizeof(uint64_t) * ctx->R * nch, hipMemcpyHostToDevice));
        ctx->spins_set = true;
        return RRRMC_OK;
    }
    // bit-sliced device layouts: upload the chunks as they are and transpose on the device (layout_kernels.hpp)
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t cbytes = sizeof(uint64_t) * (size_t)ctx->R * (size_t)nch;
    { const int32_t rci = ensure_io(ctx, cbytes); if (rci) return rci; }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(ctx->d_io, chunks, cbytes, hipMemcpyHostToDevice));
    if (ctx->model == RRRMC_MODEL_SPARSE_F64) {
        hipLaunchKernelGGL(spins_unpack_kernel<unsigned long long>, dim3((unsigned)nch, (unsigned)ctx->pfW), dim3(64), 0, ctx->stream, ctx->d_io, ctx->pf_spins, (int)N, (int)nch, (int)ctx->R);
        ctx->pf_lf_live = false;
    } else if (ctx->model == RRRMC_MODEL_SK_NORMAL || ctx->model == RRRMC_MODEL_SK_BINARY) {
        hipLaunchKernelGGL(spins_unpack_kernel<uint8_t>, dim3((unsigned)nch, (unsigned)ctx->G8), dim3(64), 0, ctx->stream, ctx->d_io, ctx->sk_spins, (int)N, (int)nch, (int)ctx->R);
    } else {
        hipLaunchKernelGGL(spins_unpack_kernel<uint32_t>, dim3((unsigned)nch, (unsigned)ctx->G), dim3(64), 0, ctx->stream, ctx->d_io, ctx->d_spins, (int)N, (int)nch, (int)ctx->R);
    }
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->spins_set = true;
    return RRRMC_OK;
}

namespace {
// the model's native device spin buffer and its size (the unit a snapshot stores)
const void* native_spins(const rrrmc_ctx* ctx)
{
    if (chunk_layout(ctx)) return ctx->q_spins;
    if (ctx->model == RRRMC_MODEL_SPARSE_F64) return ctx->pf_spins;
    if (ctx->model == RRRMC_MODEL_SK_NORMAL || ctx->model == RRRMC_MODEL_SK_BINARY) return ctx->sk_spins;
    return ctx->d_spins;
}
size_t native_spin_bytes(const rrrmc_ctx* ctx)
{
    if (chunk_layout(ctx)) return sizeof(uint32_t) * (size_t)ctx->R * (size_t)ctx->qW;
    if (ctx->model == RRRMC_MODEL_SPARSE_F64) return sizeof(unsigned long long) * (size_t)ctx->pfW * (size_t)ctx->N;
    if (ctx->model == RRRMC_MODEL_SK_NORMAL || ctx->model == RRRMC_MODEL_SK_BINARY) return (size_t)ctx->G8 * (size_t)ctx->N;
    return sizeof(uint32_t) * (size_t)ctx->G * (size_t)ctx->N;
}

// device buffer in the model's native layout -> R x ceil(N/64) BitVector chunks on the host
int32_t spins_to_chunks(rrrmc_ctx* ctx, const void* src, uint64_t* chunks)
{
    const int64_t N = ctx->N, nch = (N + 63) / 64;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (chunk_layout(ctx)) {
        HIP_TRY(ctx, hipMemcpy(chunks, src, sizeof(uint64_t) * ctx->R * nch, hipMemcpyDeviceToHost));
        return RRRMC_OK;
    }
    const size_t cbytes = sizeof(uint64_t) * (size_t)ctx->R * (size_t)nch;
    { const int32_t rci = ensure_io(ctx, cbytes); if (rci) return rci; }
    if (ctx->model == RRRMC_MODEL_SPARSE_F64)
        hipLaunchKernelGGL(spins_pack_kernel<unsigned long long>, dim3((unsigned)nch, (unsigned)ctx->pfW), dim3(64), 0, ctx->stream, (const unsigned long long*)src, ctx->d_io, (int)N, (int)nch, (int)ctx->R);
    else if (ctx->model == RRRMC_MODEL_SK_NORMAL || ctx->model == RRRMC_MODEL_SK_BINARY)
        hipLaunchKernelGGL(spins_pack_kernel<uint8_t>, dim3((unsigned)nch, (unsigned)ctx->G8), dim3(64), 0, ctx->stream, (const uint8_t*)src, ctx->d_io, (int)N, (int)nch, (int)ctx->R);
    else
        hipLaunchKernelGGL(spins_pack_kernel<uint32_t>, dim3((unsigned)nch, (unsigned)ctx->G), dim3(64), 0, ctx->stream, (const uint32_t*)src, ctx->d_io, (int)N, (int)nch, (int)ctx->R);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(chunks, ctx->d_io, cbytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return RRRMC_OK;
}
}  // namespace

int32_t rrrmc_get_spins(rrrmc_ctx* ctx, uint64_t* chunks)
{
    RRRMC_MULTI(ctx, true, rrrmc_get_spins(c, at_row(chunks, r0 * nch_of(ctx))));
    int32_t rc = ensure_state(ctx, true);
    if (rc) return rc;
    if (!chunks) return fail(ctx, RRRMC_ERR_INVALID_ARG, "chunks is NULL");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    { const int32_t rcp = post_sync_checks(ctx); if (rcp) return rcp; }          // the configuration of a call that aborted is not handed out as good
    return spins_to_chunks(ctx, native_spins(ctx), chunks);
}

int32_t rrrmc_energy(rrrmc_ctx* ctx, int64_t* E_out)
{
    RRRMC_MULTI(ctx, true, rrrmc_energy(c, at_row(E_out, r0)));
    smp_drop(ctx);
    int32_t rc = ensure_state(ctx, true);
    if (rc) return rc;
    if (!sparse_int_model(ctx)) return fail(ctx, RRRMC_ERR_STATE, "this model's energies are Float64: use the _f64 entry point");
    if (!E_out) return fail(ctx, RRRMC_ERR_INVALID_ARG, "E_out is NULL");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    rc = ctx->model == RRRMC_MODEL_SPARSE_LEVELS ? lev_standard_mc_async(ctx, 0.0, 0, 1, true) : run_energy(ctx, nullptr);
    if (rc) return rc;
    std::vector<int32_t> E((size_t)ctx->Rpad);
    HIP_TRY(ctx, hipMemcpyAsync(E.data(), ctx->d_E, sizeof(int32_t) * ctx->Rpad, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (int64_t r = 0; r < ctx->R; ++r) E_out[r] = E[r];
    return RRRMC_OK;
}

int32_t rrrmc_get_fields(rrrmc_ctx* ctx, int64_t* lfields_out)
{
    RRRMC_MULTI(ctx, true, rrrmc_get_fields(c, at_row(lfields_out, r0 * ctx->N)));
    smp_drop(ctx);
    int32_t rc = ensure_state(ctx, true);
    if (rc) return rc;
    if (ctx->model == RRRMC_MODEL_SK_BINARY) {       // the live integer cache, lfields = sqrt(N) * delta_energy (SK.jl:137-140)
        if (!lfields_out) return fail(ctx, RRRMC_ERR_INVALID_ARG, "lfields_out is NULL");
        HIP_TRY(ctx, hipSetDevice(ctx->device));
        std::vector<int32_t> lf((size_t)ctx->G8 * ctx->N * kSkRB);
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        HIP_TRY(ctx, hipMemcpy(lf.data(), ctx->skb_lf, sizeof(int32_t) * lf.size(), hipMemcpyDeviceToHost));
        for (int64_t r = 0; r < ctx->R; ++r)
            for (int64_t x = 0; x < ctx->N; ++x) lfields_out[r * ctx->N + x] = lf[((r / kSkRB) * ctx->N + x) * kSkRB + (r % kSkRB)];
        return RRRMC_OK;
    }
    if (ctx->model == RRRMC_MODEL_SPARSE_LEVELS) return fail(ctx, RRRMC_ERR_UNSUPPORTED, "the general-level kernels keep no field cache (fields are recomputed from the spins)");
    if (ctx->model != RRRMC_MODEL_SPARSE_PM1) return fail(ctx, RRRMC_ERR_STATE, "this model's energies are Float64: use the _f64 entry point");
    if (!lfields_out) return fail(ctx, RRRMC_ERR_INVALID_ARG, "lfields_out is NULL");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    uint8_t* d_nun = nullptr;
    HIP_TRY(ctx, hipMalloc(&d_nun, (size_t)ctx->Rpad * ctx->N));
    rc = run_energy(ctx, d_nun);
    if (rc) { (void)hipFree(d_nun); return rc; }
    std::vector<uint8_t> nun((size_t)ctx->Rpad * ctx->N);
    hipError_t e = hipMemcpyAsync(nun.data(), d_nun, nun.size(), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_nun);
    HIP_TRY(ctx, e);
    // lfields[x] = -dE(x) = -2 (K - 2 n)   (src/graphs/RRG.jl:164-189, 236-244)
    for (int64_t r = 0; r < ctx->R; ++r)
        for (int64_t x = 0; x < ctx->N; ++x) lfields_out[r * ctx->N + x] = 4 * (int64_t)nun[r * ctx->N + x] - 2 * ctx->K;
    return RRRMC_OK;
}

int32_t rrrmc_standard_mc_async(rrrmc_ctx* ctx, double beta, int64_t iters, int64_t step)
{
    RRRMC_MULTI(ctx, false, rrrmc_standard_mc_async(c, beta, iters, step));
    smp_drop(ctx);
    int32_t rc = ensure_state(ctx, true);
    if (rc) return rc;
    if (ctx->model == RRRMC_MODEL_SK_NORMAL || ctx->model == RRRMC_MODEL_SK_BINARY) return sk_standard_mc_async(ctx, beta, iters, step);
    if (ctx->model == RRRMC_MODEL_SPARSE_F64) return spf_standard_mc_async(ctx, beta, iters, step);
    if (ctx->model == RRRMC_MODEL_SPARSE_DISCRETIZED) return dbl_mc_async(ctx, true, beta, iters, step, 0.0, 0.0);
    if (ctx->model == RRRMC_MODEL_SPARSE_LEVELS) return lev_standard_mc_async(ctx, beta, iters, step, false);
    if (ctx->model == RRRMC_MODEL_QUANT_RRG) {
        if (!(ctx->last_fourK > 0.0)) return fail(ctx, RRRMC_ERR_STATE, "standardMC on a GraphQuant needs fourK: call rrrmc_quant_set_field first");
        return quant_mc_async(ctx, true, beta, ctx->last_fourK, iters, step, 0.0, 0.0);
    }
    if (ctx->model != RRRMC_MODEL_SPARSE_PM1) return fail(ctx, RRRMC_ERR_UNSUPPORTED, "standardMC is not wired for this model on the device: use rrrmc_rrr_mc_async");
    if (!ctx->lds_mode && !ctx->big_mode) return fail(ctx, RRRMC_ERR_UNSUPPORTED, "N=%lld is beyond the random-site kernels (N <= 2^20): use rrrmc_colored_sweeps_async", (long long)ctx->N);
    return pm1_standard_mc_async(ctx, beta, iters, step);
}

int32_t rrrmc_standard_mc_fast_async(rrrmc_ctx* ctx, double beta, int64_t iters, int64_t step)
{
    RRRMC_MULTI(ctx, false, rrrmc_standard_mc_fast_async(c, beta, iters, step));
    smp_drop(ctx);
    int32_t rc = ensure_state(ctx, true);
    if (rc) return rc;
    if (ctx->model != RRRMC_MODEL_SPARSE_F64)
        return fail(ctx, RRRMC_ERR_UNSUPPORTED, "the fast mode exists for RRRMC_MODEL_SPARSE_F64 only (the +-J models always run the bit-sliced kernel)");
    return spf_fast_mc_async(ctx, beta, iters, step);
}

int32_t rrrmc_set_resume(rrrmc_ctx* ctx, int32_t on)
{
    RRRMC_MULTI(ctx, false, rrrmc_set_resume(c, on));
    if (!ctx) return RRRMC_ERR_INVALID_ARG;
    ctx->resume = on != 0;
    return RRRMC_OK;
}

int32_t rrrmc_set_debug_checks(rrrmc_ctx* ctx, int32_t on)
{
    RRRMC_MULTI(ctx, false, rrrmc_set_debug_checks(c, on));
    if (!ctx) return RRRMC_ERR_INVALID_ARG;
    if (ctx->model != RRRMC_MODEL_SPARSE_PM1 && ctx->model != RRRMC_MODEL_SK_NORMAL && ctx->model != RRRMC_MODEL_SPARSE_F64)
        return fail(ctx, RRRMC_ERR_UNSUPPORTED, "the debug checks are wired for RRRMC_MODEL_SPARSE_PM1, RRRMC_MODEL_SK_NORMAL and RRRMC_MODEL_SPARSE_F64");
    ctx->debug_checks = on != 0;
    return RRRMC_OK;
}

int32_t rrrmc_tracked_energy_f64(rrrmc_ctx* ctx, double* E_out)
{
    RRRMC_MULTI(ctx, true, rrrmc_tracked_energy_f64(c, at_row(E_out, r0)));
    if (!ctx || !E_out) return RRRMC_ERR_INVALID_ARG;
    if (sparse_int_model(ctx)) return fail(ctx, RRRMC_ERR_STATE, "integer models: use rrrmc_tracked_energy");
    if ((!ctx->std_cache_live && ctx->smp_kind == 0) || !ctx->sk_E) return fail(ctx, RRRMC_ERR_STATE, "no sampler call has left a tracked energy");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    { const int32_t rcp = post_sync_checks(ctx); if (rcp) return rcp; }
    std::vector<double> E((size_t)ctx->Rpad);
    HIP_TRY(ctx, hipMemcpy(E.data(), ctx->sk_E, sizeof(double) * (size_t)(ctx->model == RRRMC_MODEL_QUANT_RRG || ctx->model == RRRMC_MODEL_SPARSE_DISCRETIZED ? ctx->R : ctx->Rpad), hipMemcpyDeviceToHost));
    for (int64_t r = 0; r < ctx->R; ++r) E_out[r] = E[(size_t)r];
    return RRRMC_OK;
}

// the integer models' tracked energy (level units): what the last sampler call left in E — equal to rrrmc_energy for these models, but
// read without touching the device state a resumed call continues from
int32_t rrrmc_tracked_energy(rrrmc_ctx* ctx, int64_t* E_out)
{
    RRRMC_MULTI(ctx, true, rrrmc_tracked_energy(c, at_row(E_out, r0)));
    if (!ctx || !E_out) return RRRMC_ERR_INVALID_ARG;
    if (!sparse_int_model(ctx)) return fail(ctx, RRRMC_ERR_STATE, "this model's energies are Float64: use rrrmc_tracked_energy_f64");
    if (!ctx->results_valid || !ctx->d_E) return fail(ctx, RRRMC_ERR_STATE, "no sampler call has left a tracked energy");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    { const int32_t rcp = post_sync_checks(ctx); if (rcp) return rcp; }
    std::vector<int32_t> E((size_t)ctx->Rpad);
    HIP_TRY(ctx, hipMemcpy(E.data(), ctx->d_E, sizeof(int32_t) * (size_t)ctx->Rpad, hipMemcpyDeviceToHost));
    for (int64_t r = 0; r < ctx->R; ++r) E_out[r] = E[(size_t)r];
    return RRRMC_OK;
}

int32_t rrrmc_sync(rrrmc_ctx* ctx)
{
    RRRMC_MULTI(ctx, true, rrrmc_sync(c));
    if (!ctx) return RRRMC_ERR_INVALID_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return post_sync_checks(ctx);
}

int32_t rrrmc_fetch_results(rrrmc_ctx* ctx, int64_t* Es_out, int64_t* accepted_out)
{
    RRRMC_MULTI(ctx, true, rrrmc_fetch_results(c, at_row(Es_out, r0 * c->nsamp), at_row(accepted_out, r0)));
    if (!ctx) return RRRMC_ERR_INVALID_ARG;
    if (!sparse_int_model(ctx)) return fail(ctx, RRRMC_ERR_STATE, "this model's energies are Float64: use rrrmc_fetch_results_f64");
    if (!ctx->results_valid) return fail(ctx, RRRMC_ERR_STATE, "no sampling call has been made");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    { const int32_t rcp = post_sync_checks(ctx); if (rcp) return rcp; }          // a call that aborted or failed its checks has no results to hand out
    if (accepted_out) {
        std::vector<int64_t> acc((size_t)ctx->Rpad);
        HIP_TRY(ctx, hipMemcpy(acc.data(), ctx->d_acc, sizeof(int64_t) * ctx->Rpad, hipMemcpyDeviceToHost));
        for (int64_t r = 0; r < ctx->R; ++r) accepted_out[r] = acc[r];
    }
    if (Es_out && ctx->nsamp > 0) {
        const size_t bytes = sizeof(int64_t) * (size_t)ctx->R * ctx->nsamp;
        { const int32_t rci = ensure_io(ctx, bytes); if (rci) return rci; }
        int64_t* d_out = reinterpret_cast<int64_t*>(ctx->d_io);
        const dim3 grid((unsigned)((ctx->nsamp + 31) / 32), (unsigned)ctx->G);
        hipLaunchKernelGGL(transpose_es_kernel, grid, dim3(256), 0, ctx->stream, ctx->d_Es, d_out, ctx->nsamp, (int)ctx->Rpad, (int)ctx->R);
        HIP_TRY(ctx, hipGetLastError());
        HIP_TRY(ctx, hipMemcpyAsync(Es_out, d_out, bytes, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    return RRRMC_OK;
}

int32_t rrrmc_standard_mc(rrrmc_ctx* ctx, double beta, int64_t iters, int64_t step, int64_t* Es_out, int64_t* accepted_out)
{
    int32_t rc = rrrmc_standard_mc_async(ctx, beta, iters, step);
    if (rc) return rc;
    rc = rrrmc_sync(ctx);
    if (rc) return rc;
    return rrrmc_fetch_results(ctx, Es_out, accepted_out);
}

int32_t rrrmc_last_timing(rrrmc_ctx* ctx, double* total_ms, double* sweep_ms, int32_t* sweep_launches)
{
    if (is_multi(ctx)) {          // the job's time is its slowest device's
        double tot = 0.0, sw = 0.0; int32_t nl = 0;
        for (rrrmc_ctx* c : ctx->kids) {
            double t = 0.0, w = 0.0; int32_t l = 0;
            const int32_t rc = rrrmc_last_timing(c, &t, &w, &l);
            if (rc) { ctx->err = c->err; return rc; }
            if (t > tot) tot = t;
            if (w > sw) { sw = w; nl = l; }
        }
        if (total_ms) *total_ms = tot;
        if (sweep_ms) *sweep_ms = sw;
        if (sweep_launches) *sweep_launches = nl;
        return RRRMC_OK;
    }
    if (!ctx) return RRRMC_ERR_INVALID_ARG;
    if (!ctx->timing_valid) return fail(ctx, RRRMC_ERR_STATE, "no sampling call has been made");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipEventSynchronize(ctx->ev_end));
    float ms = 0.f;
    HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev_begin, ctx->ev_end));
    if (total_ms) *total_ms = ms;
    double sw = 0.0;
    const std::vector<hipEvent_t>& EV = ctx->last_ev_pool ? ctx->ev_pool : ctx->ev_sweep;
    const size_t eb = ctx->last_ev_pool ? ctx->last_ev_base : 0;
    for (int b = 0; b < ctx->sweep_launches; ++b) {
        HIP_TRY(ctx, hipEventElapsedTime(&ms, EV[eb + 2 * b], EV[eb + 2 * b + 1]));
        sw += ms;
    }
    if (sweep_ms) *sweep_ms = sw;
    if (sweep_launches) *sweep_launches = ctx->sweep_launches;
    return RRRMC_OK;
}

int32_t rrrmc_timing_accumulate(rrrmc_ctx* ctx, int32_t on)
{
    RRRMC_MULTI(ctx, false, rrrmc_timing_accumulate(c, on));
    if (!ctx) return RRRMC_ERR_INVALID_ARG;
    if (ctx->model != RRRMC_MODEL_SPARSE_PM1) return fail(ctx, RRRMC_ERR_UNSUPPORTED, "timing accumulation is wired for the +-J sparse standardMC path only");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));      // no queued call may still record into the pool being reset
    ctx->acc_mode = on != 0;
    ctx->ev_pool_used = 0;
    for (int64_t want = 2 * (int64_t)(on > 0 ? on : 0); (int64_t)ctx->ev_pool.size() < want;) {     // on = launches to pre-create event pairs for
        hipEvent_t e;
        HIP_TRY(ctx, hipEventCreate(&e));
        ctx->ev_pool.push_back(e);
    }
    ctx->last_ev_pool = false;
    ctx->timing_valid = false;
    return RRRMC_OK;
}

int32_t rrrmc_timing_total(rrrmc_ctx* ctx, double* sweep_ms, int64_t* sweep_launches)
{
    if (is_multi(ctx)) {
        double sw = 0.0; int64_t nl = 0;
        for (rrrmc_ctx* c : ctx->kids) {
            double w = 0.0; int64_t l = 0;
            const int32_t rc = rrrmc_timing_total(c, &w, &l);
            if (rc) { ctx->err = c->err; return rc; }
            if (w > sw) { sw = w; nl = l; }
        }
        if (sweep_ms) *sweep_ms = sw;
        if (sweep_launches) *sweep_launches = nl;
        return RRRMC_OK;
    }
    if (!ctx) return RRRMC_ERR_INVALID_ARG;
    if (!ctx->acc_mode) return fail(ctx, RRRMC_ERR_STATE, "rrrmc_timing_accumulate(ctx, 1) has not been called");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    double sw = 0.0;
    for (size_t i = 0; i + 1 < ctx->ev_pool_used; i += 2) {
        float ms = 0.f;
        HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev_pool[i], ctx->ev_pool[i + 1]));
        sw += ms;
    }
    if (sweep_ms) *sweep_ms = sw;
    if (sweep_launches) *sweep_launches = (int64_t)(ctx->ev_pool_used / 2);
    return RRRMC_OK;
}

#ifdef RRRMC_STAMPS
// diagnostic builds only (tools/ablate.sh): busy shader cycles of the 16 waves of workgroup `group`, last sweep launch
RRRMC_API int32_t rrrmc_debug_stamps(rrrmc_ctx* ctx, int32_t group, unsigned long long* out16)
{
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(out16, g_stamps + (size_t)group * 16, sizeof(unsigned long long) * 16, hipMemcpyDeviceToHost));
    return RRRMC_OK;
}
// per-step busy cycles of workgroup 0: out[4096][16]
RRRMC_API int32_t rrrmc_debug_step_trace(rrrmc_ctx* ctx, unsigned long long* out)
{
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(out, g_stamps + (size_t)16 * 65536, sizeof(unsigned long long) * 16 * 4096, hipMemcpyDeviceToHost));
    return RRRMC_OK;
}
#endif

// ---- colour-parallel sweeps -----------------------------------------------------------------------------------------

int32_t rrrmc_set_coloring(rrrmc_ctx* ctx, const int32_t* color, int32_t ncolors)
{
    RRRMC_MULTI(ctx, false, rrrmc_set_coloring(c, color, ncolors));
    if (!ctx) return RRRMC_ERR_INVALID_ARG;
    if (ctx->model != RRRMC_MODEL_SPARSE_PM1) return fail(ctx, RRRMC_ERR_STATE, "colourings are for sparse +-J models");
    if (!ctx->graph_set) return fail(ctx, RRRMC_ERR_STATE, "rrrmc_set_graph has not been called");
    if (!color || ncolors < 1 || ncolors > 64) return fail(ctx, RRRMC_ERR_INVALID_ARG, "color is NULL or ncolors out of range 1..64");
    const int64_t N = ctx->N, K = ctx->K;
    std::vector<std::vector<int32_t>> lists((size_t)ncolors);
    for (int64_t x = 0; x < N; ++x) {
        if (color[x] < 0 || color[x] >= ncolors) return fail(ctx, RRRMC_ERR_INVALID_ARG, "color[%lld] = %d out of range", (long long)x, color[x]);
        for (int64_t k = 0; k < K; ++k)
            if (color[ctx->h_A[x * K + k]] == color[x])
                return fail(ctx, RRRMC_ERR_INVALID_ARG, "not a proper colouring: sites %lld and %d are adjacent and both have colour %d", (long long)x, ctx->h_A[x * K + k], color[x]);
        lists[color[x]].push_back((int32_t)x);
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (uint32_t*& l : ctx->d_color_list) free_dev(l);
    ctx->d_color_list.assign((size_t)ncolors, nullptr);
    ctx->color_count.assign((size_t)ncolors, 0);
    const int W = K <= 3 ? 4 : 8;                              // words per record (colored_sweep_kernel)
    std::vector<uint32_t> recs;
    for (int c = 0; c < ncolors; ++c) {
        ctx->color_count[c] = (int32_t)lists[c].size();
        if (lists[c].empty()) continue;
        recs.assign(lists[c].size() * (size_t)W, 0u);
        for (size_t i = 0; i < lists[c].size(); ++i) {
            const int64_t x = lists[c][i];
            recs[i * W] = (uint32_t)x;
            for (int64_t k = 0; k < K; ++k)
                recs[i * W + 1 + k] = (uint32_t)ctx->h_A[x * K + k] | (ctx->h_Jsign[x * K + k] ? 0x80000000u : 0u);
        }
        HIP_TRY(ctx, hipMalloc(&ctx->d_color_list[c], sizeof(uint32_t) * recs.size()));
        HIP_TRY(ctx, hipMemcpy(ctx->d_color_list[c], recs.data(), sizeof(uint32_t) * recs.size(), hipMemcpyHostToDevice));
    }
    ctx->ncolors = ncolors;
    return RRRMC_OK;
}

int32_t rrrmc_colored_count_accepted(rrrmc_ctx* ctx, int32_t on)
{
    RRRMC_MULTI(ctx, false, rrrmc_colored_count_accepted(c, on));
    if (!ctx) return RRRMC_ERR_INVALID_ARG;
    if (ctx->model != RRRMC_MODEL_SPARSE_PM1) return fail(ctx, RRRMC_ERR_UNSUPPORTED, "colour-parallel sweeps are for sparse +-J models");
    ctx->color_count_acc = on != 0;
    return RRRMC_OK;
}

int32_t rrrmc_colored_sweeps_async(rrrmc_ctx* ctx, double beta, int64_t sweeps, int64_t step)
{
    RRRMC_MULTI(ctx, false, rrrmc_colored_sweeps_async(c, beta, sweeps, step));
    smp_drop(ctx);
    if (ctx) ctx->std_cache_live = false;
    int32_t rc = ensure_state(ctx, true);
    if (rc) return rc;
    if (ctx->model != RRRMC_MODEL_SPARSE_PM1) return fail(ctx, RRRMC_ERR_UNSUPPORTED, "colour-parallel sweeps are for sparse +-J models");
    if (ctx->ncolors < 1) return fail(ctx, RRRMC_ERR_STATE, "rrrmc_set_coloring has not been called");
    if (sweeps < 0) return fail(ctx, RRRMC_ERR_INVALID_ARG, "sweeps must be >= 0, given %lld", (long long)sweeps);
    if (step < 1) return fail(ctx, RRRMC_ERR_INVALID_ARG, "step must be >= 1, given %lld", (long long)step);
    if (std::isnan(beta)) return fail(ctx, RRRMC_ERR_INVALID_ARG, "beta is NaN");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ctx->results_valid = false; ctx->last_call_wtm = false; ctx->last_call_eo = false;
    ctx->timing_valid = false;
    const int64_t K = ctx->K, nsamp = sweeps / step;
    const size_t es_need = (size_t)(nsamp > 0 ? nsamp : 1) * ctx->Rpad;
    if (es_need > ctx->Es_cap) {
        free_dev(ctx->d_Es);
        ctx->Es_cap = 0;
        HIP_TRY(ctx, hipMalloc(&ctx->d_Es, sizeof(int32_t) * es_need));
        ctx->Es_cap = es_need;
    }
    while (ctx->ev_sweep.size() < 2) {
        hipEvent_t e;
        HIP_TRY(ctx, hipEventCreate(&e));
        ctx->ev_sweep.push_back(e);
    }
    ColorSweepParams P{};
    const int NT = (int)(K + 1) / 2;
    for (int n = 0; n < NT; ++n) {
        bool always;
        const uint64_t T = threshold64(std::exp(-beta * 2.0 * (double)(K - 2 * n)), &always);
        if (always) P.always_mask |= 1u << n;
        for (int plane = 0; plane < 64; ++plane) P.taum[plane * 4 + n] = ((T >> (63 - plane)) & 1ull) ? ~0u : 0u;
    }
    P.spins = ctx->d_spins; P.A = ctx->d_A; P.J = ctx->d_J;
    P.k0 = (uint32_t)ctx->seed; P.k1 = (uint32_t)(ctx->seed >> 32); P.group0 = ctx->replica0 / 32; P.N = (int)ctx->N; P.G = (int)ctx->G;
    csweep_fn fn = csweep_for_K((int)K, ctx->color_count_acc);
    const int gpt = ctx->color_count_acc ? 1 : kColorGPT;
    P.acc_cur = ctx->d_acc;
    hipStream_t st = ctx->stream;
    HIP_TRY(ctx, hipEventRecord(ctx->ev_begin, st));
    // accepted counts: per replica when rrrmc_colored_count_accepted(ctx, 1) was called, else not tracked: -1
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_acc, ctx->color_count_acc ? 0 : 0xff, sizeof(int64_t) * ctx->Rpad, st));
    HIP_TRY(ctx, hipEventRecord(ctx->ev_sweep[0], st));
    int64_t ns = 0;
    for (int64_t sw = 1; sw <= sweeps; ++sw) {
        if (sw % step == 0) {      // the sample precedes sweep k*step
            rc = run_energy_bs(ctx, ctx->d_Es + ns * ctx->Rpad);
            if (rc) return rc;
            ns += 1;
        }
        P.sweep = ctx->sweeps_done + (uint64_t)sw;
        for (int c = 0; c < ctx->ncolors; ++c) {
            if (ctx->color_count[c] == 0) continue;
            P.recs = ctx->d_color_list[c];
            P.nlist = ctx->color_count[c];
            const dim3 grid((unsigned)((P.nlist + 255) / 256), (unsigned)((ctx->G + gpt - 1) / gpt));
            hipLaunchKernelGGL(fn, grid, dim3(256), 0, st, P);
            HIP_TRY(ctx, hipGetLastError());
        }
    }
    HIP_TRY(ctx, hipEventRecord(ctx->ev_sweep[1], st));
    HIP_TRY(ctx, hipEventRecord(ctx->ev_end, st));
    ctx->sweep_launches = 1;
    ctx->nsamp = nsamp;
    ctx->sweeps_done += (uint64_t)sweeps;
    ctx->results_valid = true;
    ctx->timing_valid = true;
    ctx->colored_call = true;
    return RRRMC_OK;
}

// ---- GraphQuant + rrrMC: exported entry points ---------------------------------------------------------------------

namespace {
int32_t quant_ctx_create(rrrmc_ctx** out, int64_t Nk, int64_t K, int64_t M, int64_t R, int32_t device, uint32_t replica0, bool sk, bool skn = false, bool spf = false);
}
int32_t rrrmc_ctx_create_quant(rrrmc_ctx** out, int64_t Nk, int64_t K, int64_t M, int64_t R, int32_t device, uint32_t replica0)
{
    return quant_ctx_create(out, Nk, K, M, R, device, replica0, false);
}
int32_t rrrmc_ctx_create_quant_sk(rrrmc_ctx** out, int64_t Nk, int64_t M, int64_t R, int32_t device, uint32_t replica0)
{
    return quant_ctx_create(out, Nk, 0, M, R, device, replica0, true);
}
int32_t rrrmc_ctx_create_quant_f64(rrrmc_ctx** out, int64_t Nk, int64_t K, int64_t M, int64_t R, int32_t device, uint32_t replica0)
{
    return quant_ctx_create(out, Nk, K, M, R, device, replica0, false, false, true);
}

int32_t rrrmc_ctx_create_quant_skn(rrrmc_ctx** out, int64_t Nk, int64_t M, int64_t R, int32_t device, uint32_t replica0)
{
    return quant_ctx_create(out, Nk, 0, M, R, device, replica0, false, true);
}

// One context over several devices: the reference's user makes ONE call from ONE process (src/RRRMC.jl:81-88).
// Replicas never interact, so the context is a list of per-device contexts over shards of whole 32-replica groups in global-id
// order (the random streams are addressed by global replica id: the results do not depend on ndev); every entry point forwards
// to the children — enqueue calls one after the other on each device's own stream, waiting / copying calls with one host thread
// per device — and per-replica buffers are handed over as row ranges of the caller's arrays, so results arrive gathered.
int32_t rrrmc_ctx_create_multi(rrrmc_ctx** out, int32_t model, int64_t N, int64_t K, int64_t M, int64_t R, const int32_t* device_ids,
                               int32_t ndev, uint32_t replica0)
{
    if (!out) return fail(nullptr, RRRMC_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    if (!device_ids || ndev < 1) return fail(nullptr, RRRMC_ERR_INVALID_ARG, "device_ids must name at least one device (ndev = %d)", ndev);
    if (ndev > 64) return fail(nullptr, RRRMC_ERR_INVALID_ARG, "ndev = %d: at most 64 devices", ndev);
    if (R < 1) return fail(nullptr, RRRMC_ERR_INVALID_ARG, "R must be >= 1 (given %lld)", (long long)R);
    if (replica0 % 32) return fail(nullptr, RRRMC_ERR_INVALID_ARG, "replica0 must be a multiple of 32 (given %u)", replica0);
    rrrmc_ctx* ctx = new (std::nothrow) rrrmc_ctx();
    if (!ctx) return fail(nullptr, RRRMC_ERR_NOMEM, "out of host memory");
    const bool quant = model == RRRMC_MODEL_QUANT_RRG || model == RRRMC_MODEL_QUANT_SK || model == RRRMC_MODEL_QUANT_SKN || model == RRRMC_MODEL_QUANT_F64;
    ctx->model = quant ? RRRMC_MODEL_QUANT_RRG : model; ctx->K = K; ctx->R = R; ctx->replica0 = replica0; ctx->device = device_ids[0];
    ctx->N = quant ? N * M : N;
    if (quant) { ctx->qNk = N; ctx->qM = M; ctx->q_sk = model == RRRMC_MODEL_QUANT_SK; ctx->q_skn = model == RRRMC_MODEL_QUANT_SKN; ctx->q_spf = model == RRRMC_MODEL_QUANT_F64; }
    for (int32_t d = 0; d < ndev; ++d) {
        int64_t b0 = 0, b1 = 0;
        shard_bounds(R, ndev, d, &b0, &b1);
        if (b1 <= b0) continue;          // fewer groups than devices: this one stays idle
        rrrmc_ctx* c = nullptr;
        const int32_t rc = model == RRRMC_MODEL_QUANT_RRG   ? rrrmc_ctx_create_quant(&c, N, K, M, b1 - b0, device_ids[d], replica0 + (uint32_t)b0)
                           : model == RRRMC_MODEL_QUANT_SK  ? rrrmc_ctx_create_quant_sk(&c, N, M, b1 - b0, device_ids[d], replica0 + (uint32_t)b0)
                           : model == RRRMC_MODEL_QUANT_SKN ? rrrmc_ctx_create_quant_skn(&c, N, M, b1 - b0, device_ids[d], replica0 + (uint32_t)b0)
                           : model == RRRMC_MODEL_QUANT_F64 ? rrrmc_ctx_create_quant_f64(&c, N, K, M, b1 - b0, device_ids[d], replica0 + (uint32_t)b0)
                                                            : rrrmc_ctx_create(&c, model, N, K, b1 - b0, device_ids[d], replica0 + (uint32_t)b0);
        if (rc) {
            for (rrrmc_ctx* k : ctx->kids) rrrmc_ctx_destroy(k);
            delete ctx;
            return rc;                   // (the text of the failure is already in the creation-error slot)
        }
        ctx->kids.push_back(c);
        ctx->kid_r0.push_back(b0);
    }
    *out = ctx;
    return RRRMC_OK;
}
namespace {
int32_t quant_ctx_create(rrrmc_ctx** out, int64_t Nk, int64_t K, int64_t M, int64_t R, int32_t device, uint32_t replica0, bool sk, bool skn, bool spf)
{
    if (!out) return fail(nullptr, RRRMC_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    if (Nk < 1 || (!sk && !skn && K < 1) || R < 1) return fail(nullptr, RRRMC_ERR_INVALID_ARG, "Nk, K, R must be >= 1");
    if (M <= 2) return fail(nullptr, RRRMC_ERR_INVALID_ARG, "M must be greater than 2, given: %lld", (long long)M);   // QT.jl:47
    if (spf && K > kContKmax) return fail(nullptr, RRRMC_ERR_UNSUPPORTED, "K=%lld: the Float64 sparse slice kernels cover K <= %d", (long long)K, kContKmax);
    if (Nk * M > (int64_t)1 << 28) return fail(nullptr, RRRMC_ERR_UNSUPPORTED, "N = Nk*M = %lld is beyond the GraphQuant kernels (N <= 2^28)", (long long)(Nk * M));
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, RRRMC_ERR_HIP, "no HIP device is visible: this library has no CPU path");
    if (device < 0 || device >= ndev) return fail(nullptr, RRRMC_ERR_INVALID_ARG, "device %d out of range (0..%d)", device, ndev - 1);
    rrrmc_ctx* ctx = new (std::nothrow) rrrmc_ctx();
    if (!ctx) return fail(nullptr, RRRMC_ERR_NOMEM, "out of host memory");
    ctx->model = RRRMC_MODEL_QUANT_RRG; ctx->N = Nk * M; ctx->K = K; ctx->R = R; ctx->Rpad = R;
    ctx->qNk = Nk; ctx->qM = M; ctx->qW = 2 * ((Nk * M + 63) / 64);
    ctx->q_sk = sk; ctx->q_skn = skn; ctx->q_spf = spf; ctx->q_Wk = 2 * ((Nk + 63) / 64);
    ctx->device = device; ctx->replica0 = replica0;
    const int64_t N = ctx->N;
#define Q_TRY(expr)                                                                                              \
    do {                                                                                                         \
        hipError_t e_ = (expr);                                                                                  \
        if (e_ != hipSuccess) {                                                                                  \
            int32_t rc_ = fail(nullptr, RRRMC_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));           \
            rrrmc_ctx_destroy(ctx);                                                                              \
            return rc_;                                                                                          \
        }                                                                                                        \
    } while (0)
    Q_TRY(hipSetDevice(device));
    Q_TRY(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    Q_TRY(hipEventCreate(&ctx->ev_begin));
    Q_TRY(hipEventCreate(&ctx->ev_end));
    if (skn) {
        Q_TRY(hipMalloc(&ctx->sk_J, sizeof(double) * Nk * Nk));
        Q_TRY(hipMalloc(&ctx->q_slf, sizeof(double) * (size_t)R * 2 * (size_t)M * (size_t)Nk));
        Q_TRY(hipMalloc(&ctx->q_smv, sizeof(int32_t) * (size_t)R * (size_t)M));
        Q_TRY(hipMalloc(&ctx->q_scur, (size_t)R * (size_t)M));
    } else if (sk) {
        Q_TRY(hipMalloc(&ctx->q_Jb, sizeof(uint32_t) * Nk * ctx->q_Wk));
    } else if (spf) {
        Q_TRY(hipMalloc(&ctx->d_A, sizeof(int32_t) * Nk * K));
        Q_TRY(hipMalloc(&ctx->q_Jf, sizeof(double) * Nk * K));
        Q_TRY(hipMalloc(&ctx->q_flf, sizeof(double) * (size_t)R * (size_t)M * (size_t)Nk));
        Q_TRY(hipMalloc(&ctx->q_fundo, sizeof(double) * (size_t)R * (size_t)M * (size_t)(K + 1)));
        Q_TRY(hipMalloc(&ctx->q_fml, sizeof(int32_t) * (size_t)R * (size_t)M));
    } else {
        Q_TRY(hipMalloc(&ctx->d_A, sizeof(int32_t) * Nk * K));
        Q_TRY(hipMalloc(&ctx->d_J, sizeof(int8_t) * Nk * K));
    }
    Q_TRY(hipMalloc(&ctx->q_spins, sizeof(uint32_t) * R * ctx->qW));
    Q_TRY(hipMalloc(&ctx->q_cls, (size_t)R * N));
    {
        const size_t idx_bytes = N > 65535 ? sizeof(uint32_t) : sizeof(uint16_t);      // set members / positions: 32-bit beyond 65 535 spins
        Q_TRY(hipMalloc(reinterpret_cast<void**>(&ctx->q_sv), idx_bytes * R * 4 * N));
        Q_TRY(hipMalloc(reinterpret_cast<void**>(&ctx->q_spos), idx_bytes * R * N));
    }
    Q_TRY(hipMalloc(&ctx->q_st, sizeof(int32_t) * R * 4));
    Q_TRY(hipMalloc(&ctx->q_T, sizeof(double) * R * 4));
    Q_TRY(hipMalloc(&ctx->q_z, sizeof(double) * R));
    Q_TRY(hipMalloc(&ctx->q_accrate, sizeof(double) * R));
    Q_TRY(hipMalloc(&ctx->q_stats, sizeof(int64_t) * R * 3));          // 2 per replica for rrrMC, 3 for the continuous-energy samplers
    Q_TRY(hipMalloc(&ctx->sk_E, sizeof(double) * R));
    Q_TRY(hipMemset(ctx->q_spins, 0, sizeof(uint32_t) * R * ctx->qW));
#undef Q_TRY
    *out = ctx;
    return RRRMC_OK;
}
}  // namespace

namespace {
// rrrMC(X::DoubleGraph) (standard = false) or standardMC (standard = true) on GraphQuant: thread-per-replica kernels
int32_t quant_mc_async(rrrmc_ctx* ctx, bool standard, double beta, double fourK, int64_t iters, int64_t step, double staged_thr, double staged_thr_fact)
{
    int32_t rc = RRRMC_OK;
    if (!(fourK > 0.0) || !std::isfinite(fourK)) return fail(ctx, RRRMC_ERR_INVALID_ARG, "fourK must be positive and finite, given: %g", fourK);
    if (iters < 0) return fail(ctx, RRRMC_ERR_INVALID_ARG, "iters must be >= 0, given %lld", (long long)iters);
    if (step < 1) return fail(ctx, RRRMC_ERR_INVALID_ARG, "step must be >= 1, given %lld", (long long)step);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ctx->results_valid = false; ctx->last_call_wtm = false; ctx->last_call_eo = false;
    ctx->timing_valid = false;
    // rrrMC: a resumed call finds the DeltaECache (classes, the four sets in member order, T, z), the slice caches, E and acc_rate in
    // the arrays every build of the kernel reads and writes back: it only skips energy(X, C) + gen_ΔEcache (quant_run_init)
    SmpState S{};
    if (!standard) { rc = smp_begin(ctx, 1, beta, staged_thr, staged_thr_fact, fourK, step, nullptr, &S); if (rc) return rc; }
    else S.samp0 = step;
    const int64_t nsamp = standard ? iters / step : smp_nsamp(ctx, iters, step);
    const size_t es_need = (size_t)(nsamp > 0 ? nsamp : 1) * ctx->R;
    if (es_need > ctx->sk_Es_cap) {
        free_dev(ctx->sk_Es);
        ctx->sk_Es_cap = 0;
        HIP_TRY(ctx, hipMalloc(&ctx->sk_Es, sizeof(double) * es_need));
        ctx->sk_Es_cap = es_need;
    }
    while (ctx->ev_sweep.size() < 2) {
        hipEvent_t e;
        HIP_TRY(ctx, hipEventCreate(&e));
        ctx->ev_sweep.push_back(e);
    }
    hipStream_t st = ctx->stream;
    HIP_TRY(ctx, hipEventRecord(ctx->ev_begin, st));
    ctx->last_beta = beta; ctx->last_fourK = fourK;
    ctx->stats_stride = 2;
    const bool cont = (standard && ctx->resume && ctx->std_cache_live) || S.resume;      // a resumed standardMC keeps the tracked energy (no cache)
    if (!cont) { rc = quant_run_init(ctx, beta, fourK); if (rc) return rc; }
    else if (!standard) HIP_TRY(ctx, hipMemsetAsync(ctx->q_stats, 0, sizeof(int64_t) * (size_t)ctx->R * 2, st));      // accepted / staged counts are per call
    RrrParams P = quant_params(ctx, beta, fourK);
    P.ft1 = host_det_exp(-beta * fourK);
    P.staged_thr = staged_thr;
    P.lambda = staged_thr_fact / (double)ctx->N;              // RRRMC.jl:243
    P.g0 = ctx->it_done; P.iters = iters; P.step = step; P.samp0 = S.samp0;
    HIP_TRY(ctx, hipEventRecord(ctx->ev_sweep[0], st));
    if (standard) {
        // few replicas (the reference's test_QIsing runs a handful): one wavefront per replica, 64 iterations prepared at a time
        // (quant_standard_wave_kernel); many replicas: one thread per replica fills the chip better
        const QsLayout sl = qs_layout(ctx->qW, ctx->qNk, ctx->K, ctx->q_Wk, ctx->q_sk);
        const char* no_wave = std::getenv("RRRMC_QUANT_NO_WAVE");        // timing experiments / cross-checks of the two builds
        int64_t wave_max_R = 2048;
        if (const char* e = std::getenv("RRRMC_QUANT_WAVE_MAX_R")) wave_max_R = std::atoll(e);
        const bool wave_ok = !ctx->q_skn && !ctx->q_spf && ctx->N <= 65535 && (ctx->q_sk ? ctx->qNk <= 2048 : ctx->K <= 64) && sl.bytes <= (size_t)kLdsLimit && ctx->R <= wave_max_R &&
                             !(no_wave && no_wave[0] == '1');
        if (wave_ok) {
            QsExtra X{};
            X.off_rows = (uint32_t)sl.off_rows; X.off_A = (uint32_t)sl.off_A; X.off_J = (uint32_t)sl.off_J; X.off_de = (uint32_t)sl.off_de;
            X.off_ex = (uint32_t)sl.off_ex; X.TE = sl.TE;
            if (ctx->q_sk) {
                HIP_TRY(ctx, raise_lds_attr(reinterpret_cast<const void*>(quant_standard_wave_kernel<true>), sl.bytes));
                hipLaunchKernelGGL(quant_standard_wave_kernel<true>, dim3((unsigned)ctx->R), dim3(kRrrThreads), sl.bytes, st, P, X);
            } else {
                HIP_TRY(ctx, raise_lds_attr(reinterpret_cast<const void*>(quant_standard_wave_kernel<false>), sl.bytes));
                hipLaunchKernelGGL(quant_standard_wave_kernel<false>, dim3((unsigned)ctx->R), dim3(kRrrThreads), sl.bytes, st, P, X);
            }
        } else {
            hipLaunchKernelGGL(quant_standard_kernel, dim3(rrr_blocks(ctx->R)), dim3(rrr_tpb(ctx->R)), 0, st, P);
        }
    } else {
        // Few replicas of a cache that fits LDS (config 5: 128 x (Nk = 1024, M = 32)): one WAVEFRONT per replica, wave-uniform chain,
        // the whole DeltaECache in LDS (quant_wave_kernel.hpp).  Many replicas: the thread-per-replica kernels fill the chip better.
        QwLayout ql = qw_layout(ctx->N, ctx->qW, ctx->qNk, ctx->K, (size_t)kLdsLimit, ctx->q_sk);
        if (const char* e = std::getenv("RRRMC_QUANT_WAVE_SLACK")) {      // tests: a small slack makes the segments re-space often
            const int64_t want = ctx->N + std::atoll(e);
            if (want >= ctx->N + 4 * kQwMinGap && want < ql.cap) ql.cap = (int)(want & ~(int64_t)3);
        }
        const char* no_wave = std::getenv("RRRMC_QUANT_NO_WAVE");        // timing experiments / cross-checks of the two builds
        int64_t wave_max_R = 2048;
        if (const char* e = std::getenv("RRRMC_QUANT_WAVE_MAX_R")) wave_max_R = std::atoll(e);
        // GraphRRG / GraphEA slices with K <= 7, or binary GraphSK slices of up to 2048 spins (one word of the slice per lane)
        const bool wave_ok = !ctx->q_skn && !ctx->q_spf && ctx->N <= 65535 && (ctx->q_sk ? ctx->qNk <= 2048 : ctx->K <= 7) && ql.cap >= ctx->N + 4 * kQwMinGap &&
                             ctx->R <= wave_max_R && !(no_wave && no_wave[0] == '1');
        // one replica per workgroup anyway (few replicas): stage its hot state in LDS if it fits (config 5: 115 KB)
        const size_t lds = rrr_quant_lds_bytes(ctx->N, ctx->qW, ctx->qNk, ctx->K);
        const char* no_lds = std::getenv("RRRMC_QUANT_NO_LDS");          // timing experiments
        if (wave_ok) {
            HIP_TRY(ctx, raise_lds_attr(ctx->q_sk ? reinterpret_cast<const void*>(rrr_quant_wave_kernel<true>) : reinterpret_cast<const void*>(rrr_quant_wave_kernel<false>), ql.bytes));
            QwExtra X{};
            X.cap = ql.cap; X.off_spos = (uint32_t)ql.off_spos; X.off_sv = (uint32_t)ql.off_sv; X.off_A = (uint32_t)ql.off_A;
            X.off_J = (uint32_t)ql.off_J; X.off_rng = (uint32_t)ql.off_rng; X.off_tab = (uint32_t)ql.off_tab;
#ifdef RRRMC_STAMPS
            if (!g_stamps) HIP_TRY(ctx, hipMalloc(&g_stamps, sizeof(unsigned long long) * 16 * (65536 + 4096)));
            X.stamps = g_stamps;
#endif
            if (ctx->q_sk) hipLaunchKernelGGL(rrr_quant_wave_kernel<true>, dim3((unsigned)ctx->R), dim3(kRrrThreads), ql.bytes, st, P, X);
            else hipLaunchKernelGGL(rrr_quant_wave_kernel<false>, dim3((unsigned)ctx->R), dim3(kRrrThreads), ql.bytes, st, P, X);
        } else if (!ctx->q_skn && !ctx->q_spf && ctx->N <= 65535 && rrr_tpb(ctx->R) == 1 && lds <= (size_t)kLdsLimit && !(no_lds && no_lds[0] == '1')) {
            if (!ctx->q_lds_attr) {
                HIP_TRY(ctx, raise_lds_attr(reinterpret_cast<const void*>(rrr_quant_kernel<true>), lds));
                ctx->q_lds_attr = true;
            }
            hipLaunchKernelGGL(rrr_quant_kernel<true>, dim3((unsigned)ctx->R), dim3(kRrrThreads), lds, st, P);
        } else {
            hipLaunchKernelGGL(rrr_quant_kernel<false>, dim3(rrr_blocks(ctx->R)), dim3(rrr_tpb(ctx->R)), 0, st, P);
        }
    }
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipEventRecord(ctx->ev_sweep[1], st));
    HIP_TRY(ctx, hipEventRecord(ctx->ev_end, st));
    ctx->sweep_launches = 1;
    ctx->nsamp = nsamp;
    ctx->it_done += (uint64_t)iters;
    if (!standard) smp_commit(ctx, 1, iters);
    ctx->results_valid = true;
    ctx->timing_valid = true;
    ctx->last_call_rrr = true;          // accepted counts live in q_stats
    ctx->q_cache_valid = !standard;
    ctx->std_cache_live = standard;
    return RRRMC_OK;
}
}  // namespace

int32_t rrrmc_rrr_mc_async(rrrmc_ctx* ctx, double beta, double fourK, int64_t iters, int64_t step, double staged_thr, double staged_thr_fact)
{
    RRRMC_MULTI(ctx, false, rrrmc_rrr_mc_async(c, beta, fourK, iters, step, staged_thr, staged_thr_fact));
    if (ctx) ctx->std_cache_live = false;
    int32_t rc = ensure_state(ctx, true);
    if (rc) return rc;
    if (!std::isfinite(beta)) return fail(ctx, RRRMC_ERR_INVALID_ARG, "beta must be finite, given: %g", beta);       // RRRMC.jl:230, :166
    if (ctx->model == RRRMC_MODEL_SK_NORMAL || ctx->model == RRRMC_MODEL_SK_BINARY) return sk_rrr_mc_async(ctx, beta, iters, step, staged_thr, staged_thr_fact);
    if (sparse_int_model(ctx)) return sparse_rrr_bkl_async(ctx, 0, beta, iters, step, staged_thr, staged_thr_fact);
    if (ctx->model == RRRMC_MODEL_SPARSE_DISCRETIZED) return dbl_rrr_mc_async(ctx, beta, iters, step, staged_thr, staged_thr_fact);
    if (ctx->model == RRRMC_MODEL_SPARSE_F64) return spf_cont_async(ctx, 0, beta, iters, step, 1.0, staged_thr, staged_thr_fact);
    if (ctx->model != RRRMC_MODEL_QUANT_RRG) return fail(ctx, RRRMC_ERR_UNSUPPORTED, "rrrMC is not available for model kind %d", ctx->model);
    return quant_mc_async(ctx, false, beta, fourK, iters, step, staged_thr, staged_thr_fact);
}

int32_t rrrmc_bkl_mc_async(rrrmc_ctx* ctx, double beta, int64_t iters, int64_t step)
{
    RRRMC_MULTI(ctx, false, rrrmc_bkl_mc_async(c, beta, iters, step));
    if (ctx) ctx->std_cache_live = false;
    int32_t rc = ensure_state(ctx, true);
    if (rc) return rc;
    if (!std::isfinite(beta)) return fail(ctx, RRRMC_ERR_INVALID_ARG, "beta must be finite, given: %g", beta);
    if (ctx->model == RRRMC_MODEL_SPARSE_F64 || ctx->model == RRRMC_MODEL_SPARSE_DISCRETIZED || ctx->model == RRRMC_MODEL_QUANT_RRG)
        return spf_cont_async(ctx, 1, beta, iters, step, 1.0, 0.0, 5.0);
    if (ctx->model == RRRMC_MODEL_SK_NORMAL || ctx->model == RRRMC_MODEL_SK_BINARY) return sk_rrr_mc_async(ctx, beta, iters, step, 0.0, 5.0, 1);
    if (!sparse_int_model(ctx)) return fail(ctx, RRRMC_ERR_UNSUPPORTED, "bklMC is wired for the sparse models, RRRMC_MODEL_SK_NORMAL and RRRMC_MODEL_SK_BINARY");
    return sparse_rrr_bkl_async(ctx, 1, beta, iters, step, 0.0, 5.0);
}

int32_t rrrmc_quant_slice_form(rrrmc_ctx* ctx, int32_t ea_form)
{
    RRRMC_MULTI(ctx, false, rrrmc_quant_slice_form(c, ea_form));
    smp_drop(ctx);
    if (!ctx) return RRRMC_ERR_INVALID_ARG;
    if (ctx->model != RRRMC_MODEL_QUANT_RRG || ctx->q_sk) return fail(ctx, RRRMC_ERR_STATE, "rrrmc_quant_slice_form is for a GraphQuant over GraphRRG / GraphEA slices");
    if (ctx->graph_set) return fail(ctx, RRRMC_ERR_STATE, "rrrmc_quant_slice_form must precede rrrmc_set_graph");
    ctx->db_ea_form = ea_form ? 1 : 0;
    return RRRMC_OK;
}

int32_t rrrmc_quant_set_field(rrrmc_ctx* ctx, double beta, double fourK)
{
    RRRMC_MULTI(ctx, false, rrrmc_quant_set_field(c, beta, fourK));
    smp_drop(ctx);
    if (!ctx) return RRRMC_ERR_INVALID_ARG;
    if (ctx->model != RRRMC_MODEL_QUANT_RRG) return fail(ctx, RRRMC_ERR_STATE, "rrrmc_quant_set_field is for RRRMC_MODEL_QUANT_RRG");
    if (!(fourK > 0.0) || !std::isfinite(fourK) || !std::isfinite(beta)) return fail(ctx, RRRMC_ERR_INVALID_ARG, "beta and fourK must be finite, fourK > 0");
    ctx->last_beta = beta;
    ctx->last_fourK = fourK;
    return RRRMC_OK;
}

int32_t rrrmc_wtm_mc_async(rrrmc_ctx* ctx, double beta, int64_t samples, double step)
{
    RRRMC_MULTI(ctx, false, rrrmc_wtm_mc_async(c, beta, samples, step));
    if (ctx) ctx->std_cache_live = false;
    int32_t rc = ensure_state(ctx, true);
    if (rc) return rc;
    if (ctx->model == RRRMC_MODEL_SPARSE_F64 || ctx->model == RRRMC_MODEL_SPARSE_DISCRETIZED || ctx->model == RRRMC_MODEL_QUANT_RRG)
        return spf_cont_async(ctx, 2, beta, samples, 1, step, 0.0, 5.0);
    if (ctx->model == RRRMC_MODEL_SK_NORMAL || ctx->model == RRRMC_MODEL_SK_BINARY) return sk_rrr_mc_async(ctx, beta, samples, 1, 0.0, 5.0, 2, step);
    if (!sparse_int_model(ctx)) return fail(ctx, RRRMC_ERR_UNSUPPORTED, "wtmMC is wired for the sparse models, RRRMC_MODEL_SK_NORMAL and RRRMC_MODEL_SK_BINARY");
    return sparse_wtm_async(ctx, beta, samples, step);
}

int32_t rrrmc_wtm_times(rrrmc_ctx* ctx, double* t_out)
{
    RRRMC_MULTI(ctx, true, rrrmc_wtm_times(c, at_row(t_out, r0)));
    if (!ctx) return RRRMC_ERR_INVALID_ARG;
    if (!ctx->last_call_wtm || !ctx->results_valid) return fail(ctx, RRRMC_ERR_STATE, "no wtmMC call has been made");
    if (!t_out) return fail(ctx, RRRMC_ERR_INVALID_ARG, "t_out is NULL");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(t_out, ctx->wt_time, sizeof(double) * ctx->R, hipMemcpyDeviceToHost));
    return RRRMC_OK;
}

int32_t rrrmc_extremal_opt_async(rrrmc_ctx* ctx, const double* ftau, int64_t iters, int64_t step)
{
    RRRMC_MULTI(ctx, false, rrrmc_extremal_opt_async(c, ftau, iters, step));
    if (ctx) ctx->std_cache_live = false;
    int32_t rc = ensure_state(ctx, true);
    if (rc) return rc;
    if (ctx->model == RRRMC_MODEL_SPARSE_F64 || ctx->model == RRRMC_MODEL_SPARSE_DISCRETIZED || ctx->model == RRRMC_MODEL_QUANT_RRG)        // not DiscrGraphs: EOCacheCont
        return spf_cont_async(ctx, 3, 0.0, iters, step, 1.0, 0.0, 0.0, ftau);
    if (ctx->model == RRRMC_MODEL_SK_NORMAL || ctx->model == RRRMC_MODEL_SK_BINARY)                 // EOCacheCont, every spin a neighbour
        return sk_rrr_mc_async(ctx, 0.0, iters, step, 0.0, 5.0, 3, 1.0, ftau);
    if (!sparse_int_model(ctx)) return fail(ctx, RRRMC_ERR_UNSUPPORTED, "extremal_opt is not wired for this model");
    return sparse_eo_async(ctx, ftau, iters, step);
}

namespace {
// shared by the integer and the Float64 variants: Emin as int64 (q_stats[.][0]) or double (wt_time), never both
int32_t eo_results(rrrmc_ctx* ctx, int64_t* Emin_i, double* Emin_f, uint64_t* Cmin_chunks, int64_t* itmin_out)
{
    if (!ctx->last_call_eo || !ctx->results_valid) return fail(ctx, RRRMC_ERR_STATE, "no extremal_opt call has been made");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    const int64_t ss = ctx->stats_stride;
    std::vector<int64_t> st((size_t)ctx->R * (size_t)ss);
    HIP_TRY(ctx, hipMemcpy(st.data(), ctx->q_stats, sizeof(int64_t) * st.size(), hipMemcpyDeviceToHost));
    for (int64_t r = 0; r < ctx->R; ++r) {
        if (Emin_i) Emin_i[r] = st[(size_t)(ss * r)];
        if (itmin_out) itmin_out[r] = st[(size_t)(ss * r + 1)];
    }
    if (Emin_f) HIP_TRY(ctx, hipMemcpy(Emin_f, ctx->wt_time, sizeof(double) * ctx->R, hipMemcpyDeviceToHost));
    if (Cmin_chunks) {
        const int64_t nch = (ctx->N + 63) / 64, W = ctx->eo_W;
        std::vector<uint32_t> w((size_t)ctx->R * (size_t)W);
        HIP_TRY(ctx, hipMemcpy(w.data(), ctx->eo_cmin, sizeof(uint32_t) * w.size(), hipMemcpyDeviceToHost));
        std::memset(Cmin_chunks, 0, sizeof(uint64_t) * ctx->R * nch);
        for (int64_t r = 0; r < ctx->R; ++r)
            for (int64_t q = 0; q < W; ++q) Cmin_chunks[r * nch + (q >> 1)] |= (uint64_t)w[(size_t)(r * W + q)] << (32 * (q & 1));
    }
    return RRRMC_OK;
}
}  // namespace

int32_t rrrmc_extremal_opt_results(rrrmc_ctx* ctx, int64_t* Emin_out, uint64_t* Cmin_chunks, int64_t* itmin_out)
{
    RRRMC_MULTI(ctx, true, rrrmc_extremal_opt_results(c, at_row(Emin_out, r0), at_row(Cmin_chunks, r0 * nch_of(ctx)), at_row(itmin_out, r0)));
    if (!ctx) return RRRMC_ERR_INVALID_ARG;
    if (!sparse_int_model(ctx)) return fail(ctx, RRRMC_ERR_STATE, "this model's energies are Float64: use rrrmc_extremal_opt_results_f64");
    return eo_results(ctx, Emin_out, nullptr, Cmin_chunks, itmin_out);
}

int32_t rrrmc_extremal_opt_results_f64(rrrmc_ctx* ctx, double* Emin_out, uint64_t* Cmin_chunks, int64_t* itmin_out)
{
    RRRMC_MULTI(ctx, true, rrrmc_extremal_opt_results_f64(c, at_row(Emin_out, r0), at_row(Cmin_chunks, r0 * nch_of(ctx)), at_row(itmin_out, r0)));
    if (!ctx) return RRRMC_ERR_INVALID_ARG;
    if (sparse_int_model(ctx)) return fail(ctx, RRRMC_ERR_STATE, "this model's energies are integers: use rrrmc_extremal_opt_results");
    return eo_results(ctx, nullptr, Emin_out, Cmin_chunks, itmin_out);
}

int32_t rrrmc_rrr_stats(rrrmc_ctx* ctx, int64_t* staged_iters_out)
{
    RRRMC_MULTI(ctx, true, rrrmc_rrr_stats(c, at_row(staged_iters_out, r0)));
    if (!ctx) return RRRMC_ERR_INVALID_ARG;
    if (!ctx->last_call_rrr || !ctx->results_valid) return fail(ctx, RRRMC_ERR_STATE, "no rrrMC call has been made");
    if (!staged_iters_out) return fail(ctx, RRRMC_ERR_INVALID_ARG, "staged_iters_out is NULL");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<int64_t> st((size_t)ctx->R * ctx->stats_stride);
    HIP_TRY(ctx, hipMemcpy(st.data(), ctx->q_stats, sizeof(int64_t) * st.size(), hipMemcpyDeviceToHost));
    for (int64_t r = 0; r < ctx->R; ++r) staged_iters_out[r] = st[ctx->stats_stride * r + 1];
    return RRRMC_OK;
}

int32_t rrrmc_rrr_cache(rrrmc_ctx* ctx, int8_t* pos_out, int32_t* sizes_out)
{
    RRRMC_MULTI(ctx, true, rrrmc_rrr_cache(c, at_row(pos_out, r0 * ctx->N), at_row(sizes_out, r0 * (ctx->model == RRRMC_MODEL_QUANT_RRG ? 4 : 16))));
    if (!ctx) return RRRMC_ERR_INVALID_ARG;
    if ((ctx->model != RRRMC_MODEL_QUANT_RRG && ctx->model != RRRMC_MODEL_SPARSE_DISCRETIZED && !sparse_int_model(ctx)) || !ctx->results_valid || !ctx->last_call_rrr)
        return fail(ctx, RRRMC_ERR_STATE, "no rrrMC call has been made");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (sparse_int_model(ctx)) {          // DeltaECache{Int,L} of rrrMC / bklMC on GraphRRG / GraphEA: classes from the kernels' class bytes, sizes from the run's slab
        if ((ctx->smp_kind != 1 && ctx->smp_kind != 2) || !ctx->rp_cls) return fail(ctx, RRRMC_ERR_STATE, "no rrrMC / bklMC call has been made");
        if (pos_out) HIP_TRY(ctx, hipMemcpy(pos_out, ctx->rp_cls, (size_t)ctx->R * ctx->N, hipMemcpyDeviceToHost));
        if (sizes_out) {
            std::vector<long long> si((size_t)ctx->R * kSmpI);
            HIP_TRY(ctx, hipMemcpy(si.data(), ctx->smp_i, sizeof(long long) * si.size(), hipMemcpyDeviceToHost));
            for (int64_t r = 0; r < ctx->R; ++r)
                for (int k = 0; k < 16; ++k) sizes_out[r * 16 + k] = k < 2 * ctx->lv.L ? (int32_t)si[(size_t)r * kSmpI + SI_T0 + k] : 0;
        }
        return RRRMC_OK;
    }
    if (ctx->model == RRRMC_MODEL_QUANT_RRG && !ctx->q_cache_valid) return fail(ctx, RRRMC_ERR_STATE, "no rrrMC call has been made");
    if (ctx->model == RRRMC_MODEL_SPARSE_DISCRETIZED) {      // sizes_out[R * 16]: counted from the classes
        if (!ctx->db_cache_valid) return fail(ctx, RRRMC_ERR_STATE, "no rrrMC call has been made");
        std::vector<uint8_t> cls((size_t)ctx->R * ctx->N);
        HIP_TRY(ctx, hipMemcpy(cls.data(), ctx->db_cls, cls.size(), hipMemcpyDeviceToHost));
        const int K2 = 2 * kDLmax;                           // fixed stride: class k of replica r at sizes_out[16 r + k]
        if (pos_out) std::memcpy(pos_out, cls.data(), cls.size());
        if (sizes_out) {
            std::memset(sizes_out, 0, sizeof(int32_t) * (size_t)ctx->R * K2);
            for (int64_t r = 0; r < ctx->R; ++r)
                for (int64_t i = 0; i < ctx->N; ++i) sizes_out[r * K2 + cls[(size_t)(r * ctx->N + i)]] += 1;
        }
        return RRRMC_OK;
    }
    if (pos_out) HIP_TRY(ctx, hipMemcpy(pos_out, ctx->q_cls, (size_t)ctx->R * ctx->N, hipMemcpyDeviceToHost));
    if (sizes_out) HIP_TRY(ctx, hipMemcpy(sizes_out, ctx->q_st, sizeof(int32_t) * ctx->R * 4, hipMemcpyDeviceToHost));
    return RRRMC_OK;
}

// ---- Float64-energy models: exported entry points ---------------------------------------------------------------

int32_t rrrmc_set_couplings_dense(rrrmc_ctx* ctx, const double* J)
{
    RRRMC_MULTI(ctx, false, rrrmc_set_couplings_dense(c, J));
    smp_drop(ctx);
    if (!ctx) return RRRMC_ERR_INVALID_ARG;
    const bool qskn = ctx->model == RRRMC_MODEL_QUANT_RRG && ctx->q_skn;      // the slice graph of a GraphQSKNormalT: J is Nk x Nk
    if (ctx->model != RRRMC_MODEL_SK_NORMAL && !qskn) return fail(ctx, RRRMC_ERR_STATE, "rrrmc_set_couplings_dense is for RRRMC_MODEL_SK_NORMAL (or a GraphQuant over GraphSKNormal slices)");
    if (!J) return fail(ctx, RRRMC_ERR_INVALID_ARG, "J is NULL");
    const int64_t N = qskn ? ctx->qNk : ctx->N;
    for (int64_t i = 0; i < N; ++i) {        // GraphSKNormal(J; check = true), SK.jl:187-194
        if (J[i * N + i] != 0.0) return fail(ctx, RRRMC_ERR_INVALID_ARG, "diagonal entries of J must be 0, found: J[%lld][%lld] = %g", (long long)i, (long long)i, J[i * N + i]);
        for (int64_t j = i + 1; j < N; ++j)
            if (!(J[i * N + j] == J[j * N + i]))
                return fail(ctx, RRRMC_ERR_INVALID_ARG, "J must be symmetric, found: J[%lld][%lld] = %g, J[%lld][%lld] = %g", (long long)i, (long long)j, J[i * N + j], (long long)j, (long long)i, J[j * N + i]);
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(ctx->sk_J, J, sizeof(double) * N * N, hipMemcpyHostToDevice));
    if (ctx->sk_J4) {          // 4 J, exact: the increment of update_cache! (SK.jl:256-262) as the blocked standardMC kernel adds it
        const int64_t ld = sk_ldJ(N);
        std::vector<double> J4((size_t)(N * ld), 0.0);
        for (int64_t i = 0; i < N; ++i)
            for (int64_t j = 0; j < N; ++j) J4[(size_t)(i * ld + j)] = 4.0 * J[i * N + j];
        HIP_TRY(ctx, hipMemcpy(ctx->sk_J4, J4.data(), sizeof(double) * N * ld, hipMemcpyHostToDevice));
    }
    ctx->graph_set = true;
    return RRRMC_OK;
}

int32_t rrrmc_energy_f64(rrrmc_ctx* ctx, double* E_out)
{
    RRRMC_MULTI(ctx, true, rrrmc_energy_f64(c, at_row(E_out, r0)));
    smp_drop(ctx);
    int32_t rc = ensure_state(ctx, true);
    if (rc) return rc;
    if (sparse_int_model(ctx)) return fail(ctx, RRRMC_ERR_STATE, "this model's energies are integers: use rrrmc_energy");
    if (!E_out) return fail(ctx, RRRMC_ERR_INVALID_ARG, "E_out is NULL");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (ctx->model == RRRMC_MODEL_QUANT_RRG) {
        if (!(ctx->last_fourK > 0.0)) return fail(ctx, RRRMC_ERR_STATE, "energy of a GraphQuant needs fourK: call rrrmc_quant_set_field first");
        rc = quant_run_init(ctx, ctx->last_beta, ctx->last_fourK);
    } else if (ctx->model == RRRMC_MODEL_SPARSE_F64) {
        rc = spf_run_energy(ctx);
    } else if (ctx->model == RRRMC_MODEL_SPARSE_DISCRETIZED) {
        rc = dbl_run_energy(ctx);
        // (this evaluation leaves the residual fields as they are and overwrites the tracked energy: what a resumed standardMC would find is
        // neither the run's state nor a fresh one — the next call starts fresh, as after the other models' energy calls, which rebuild in place)
        ctx->std_cache_live = false;
    } else {
        rc = sk_run_energy(ctx);
    }
    if (rc) return rc;
    std::vector<double> E((size_t)ctx->Rpad);
    HIP_TRY(ctx, hipMemcpyAsync(E.data(), ctx->sk_E, sizeof(double) * ctx->Rpad, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (int64_t r = 0; r < ctx->R; ++r) E_out[r] = E[r];
    return RRRMC_OK;
}

int32_t rrrmc_get_fields_f64(rrrmc_ctx* ctx, double* lfields_out)
{
    RRRMC_MULTI(ctx, true, rrrmc_get_fields_f64(c, at_row(lfields_out, r0 * ctx->N)));
    int32_t rc = ensure_state(ctx, true);
    if (rc) return rc;
    if (ctx->model != RRRMC_MODEL_SK_NORMAL && ctx->model != RRRMC_MODEL_SPARSE_F64)
        return fail(ctx, RRRMC_ERR_STATE, "this model's fields are integers (or not cached): use rrrmc_get_fields");
    if (!lfields_out) return fail(ctx, RRRMC_ERR_INVALID_ARG, "lfields_out is NULL");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int64_t N = ctx->N;
    if (ctx->model == RRRMC_MODEL_SPARSE_F64) {
        if (!ctx->pf_lf_live) return fail(ctx, RRRMC_ERR_STATE, "the cached local fields are not current (the last sampler kept its own): call rrrmc_energy_f64 first");
        std::vector<double> lf((size_t)ctx->Rpad * N);
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        HIP_TRY(ctx, hipMemcpy(lf.data(), ctx->sk_lf, sizeof(double) * lf.size(), hipMemcpyDeviceToHost));
        for (int64_t r = 0; r < ctx->R; ++r)
            for (int64_t x = 0; x < N; ++x) lfields_out[r * N + x] = lf[((r >> 6) * N + x) * 64 + (r & 63)];
        return RRRMC_OK;
    }
    std::vector<double> lf((size_t)ctx->G8 * N * kSkRB);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(lf.data(), ctx->sk_lf, sizeof(double) * lf.size(), hipMemcpyDeviceToHost));   // the live cache (not recomputed)
    for (int64_t r = 0; r < ctx->R; ++r)
        for (int64_t x = 0; x < N; ++x) lfields_out[r * N + x] = lf[((r / kSkRB) * N + x) * kSkRB + (r % kSkRB)];
    return RRRMC_OK;
}

int32_t rrrmc_fetch_results_f64(rrrmc_ctx* ctx, double* Es_out, int64_t* accepted_out)
{
    RRRMC_MULTI(ctx, true, rrrmc_fetch_results_f64(c, at_row(Es_out, r0 * c->nsamp), at_row(accepted_out, r0)));
    if (!ctx) return RRRMC_ERR_INVALID_ARG;
    if (sparse_int_model(ctx)) return fail(ctx, RRRMC_ERR_STATE, "this model's energies are integers: use rrrmc_fetch_results");
    if (!ctx->results_valid) return fail(ctx, RRRMC_ERR_STATE, "no sampling call has been made");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    { const int32_t rcp = post_sync_checks(ctx); if (rcp) return rcp; }          // a call that aborted or failed its checks has no results to hand out
    if (accepted_out && ctx->last_call_rrr) {
        std::vector<int64_t> st((size_t)ctx->R * ctx->stats_stride);
        HIP_TRY(ctx, hipMemcpy(st.data(), ctx->q_stats, sizeof(int64_t) * st.size(), hipMemcpyDeviceToHost));
        for (int64_t r = 0; r < ctx->R; ++r) accepted_out[r] = st[ctx->stats_stride * r];
    } else if (accepted_out) {
        std::vector<int64_t> acc((size_t)ctx->Rpad);
        HIP_TRY(ctx, hipMemcpy(acc.data(), ctx->d_acc, sizeof(int64_t) * ctx->Rpad, hipMemcpyDeviceToHost));
        for (int64_t r = 0; r < ctx->R; ++r) accepted_out[r] = acc[r];
    }
    if (Es_out && ctx->nsamp > 0) {
        double* d_out = nullptr;
        const size_t bytes = sizeof(double) * (size_t)ctx->R * ctx->nsamp;
        HIP_TRY(ctx, hipMalloc(&d_out, bytes));
        const dim3 grid((unsigned)((ctx->nsamp + 31) / 32), (unsigned)((ctx->Rpad + 31) / 32));
        hipLaunchKernelGGL(transpose_es_f64_kernel, grid, dim3(256), 0, ctx->stream, ctx->sk_Es, d_out, ctx->nsamp, (int)ctx->Rpad, (int)ctx->R);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(Es_out, d_out, bytes, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        (void)hipFree(d_out);
        HIP_TRY(ctx, e);
    }
    return RRRMC_OK;
}

int32_t rrrmc_standard_mc_f64(rrrmc_ctx* ctx, double beta, int64_t iters, int64_t step, double* Es_out, int64_t* accepted_out)
{
    int32_t rc = rrrmc_standard_mc_async(ctx, beta, iters, step);
    if (rc) return rc;
    rc = rrrmc_sync(ctx);
    if (rc) return rc;
    return rrrmc_fetch_results_f64(ctx, Es_out, accepted_out);
}

int32_t rrrmc_set_couplings_bits(rrrmc_ctx* ctx, const uint64_t* Jc)
{
    RRRMC_MULTI(ctx, false, rrrmc_set_couplings_bits(c, Jc));
    smp_drop(ctx);
    if (!ctx) return RRRMC_ERR_INVALID_ARG;
    const bool qsk = ctx->model == RRRMC_MODEL_QUANT_RRG && ctx->q_sk;          // the slice graph of a GraphQSKT
    if (ctx->model != RRRMC_MODEL_SK_BINARY && !qsk)
        return fail(ctx, RRRMC_ERR_STATE, "rrrmc_set_couplings_bits is for RRRMC_MODEL_SK_BINARY and for contexts made by rrrmc_ctx_create_quant_sk");
    if (!Jc) return fail(ctx, RRRMC_ERR_INVALID_ARG, "J_chunks is NULL");
    const int64_t N = qsk ? ctx->qNk : ctx->N, nch = (N + 63) / 64;
    auto bit = [&](int64_t i, int64_t j) { return (int)((Jc[i * nch + (j >> 6)] >> (j & 63)) & 1ull); };
    for (int64_t i = 0; i < N; ++i) {        // GraphSK(J; check = true), SK.jl:36-46
        if (bit(i, i)) return fail(ctx, RRRMC_ERR_INVALID_ARG, "diagonal entries of J must be 0, found: J[%lld][%lld] = 1", (long long)i, (long long)i);
        for (int64_t j = i + 1; j < N; ++j)
            if (bit(i, j) != bit(j, i)) return fail(ctx, RRRMC_ERR_INVALID_ARG, "J must be symmetric, found: J[%lld][%lld] != J[%lld][%lld]", (long long)i, (long long)j, (long long)j, (long long)i);
        if (N % 64 && (Jc[i * nch + nch - 1] >> (N % 64))) return fail(ctx, RRRMC_ERR_INVALID_ARG, "row %lld: bits beyond N are set", (long long)i);
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(qsk ? (void*)ctx->q_Jb : (void*)ctx->skb_J, Jc, sizeof(uint64_t) * N * nch, hipMemcpyHostToDevice));     // chunk = two little-endian words
    if (qsk) ctx->q_cache_valid = false;
    ctx->skb_J4_valid = false;
    ctx->graph_set = true;
    return RRRMC_OK;
}

// ---------------------------------------------------------------------------------------------------
// Snapshots and observables
// ---------------------------------------------------------------------------------------------------
int32_t rrrmc_snapshot_reserve(rrrmc_ctx* ctx, int32_t nslots)
{
    RRRMC_MULTI(ctx, false, rrrmc_snapshot_reserve(c, nslots));
    if (!ctx) return RRRMC_ERR_INVALID_ARG;
    if (nslots < 0) return fail(ctx, RRRMC_ERR_INVALID_ARG, "nslots must be >= 0");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    free_dev(ctx->snap);
    ctx->snap_slots = 0;
    ctx->snap_valid.clear();
    if (nslots == 0) return RRRMC_OK;
    const size_t bytes = native_spin_bytes(ctx) * (size_t)nslots;
    if (hipMalloc(&ctx->snap, bytes) != hipSuccess) {
        ctx->snap = nullptr;
        return fail(ctx, RRRMC_ERR_NOMEM, "cannot allocate %zu bytes for %d snapshots", bytes, nslots);
    }
    ctx->snap_slots = nslots;
    ctx->snap_valid.assign((size_t)nslots, 0);
    return RRRMC_OK;
}

int32_t rrrmc_snapshot_store(rrrmc_ctx* ctx, int32_t slot)
{
    RRRMC_MULTI(ctx, false, rrrmc_snapshot_store(c, slot));
    int32_t rc = ensure_state(ctx, true);
    if (rc) return rc;
    if (slot < 0 || slot >= ctx->snap_slots) return fail(ctx, RRRMC_ERR_INVALID_ARG, "snapshot slot %d out of range (0..%d)", slot, ctx->snap_slots - 1);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t sb = native_spin_bytes(ctx);
    HIP_TRY(ctx, hipMemcpyAsync(ctx->snap + sb * (size_t)slot, native_spins(ctx), sb, hipMemcpyDeviceToDevice, ctx->stream));
    ctx->snap_valid[(size_t)slot] = 1;
    return RRRMC_OK;
}

int32_t rrrmc_snapshot_get(rrrmc_ctx* ctx, int32_t slot, uint64_t* chunks)
{
    RRRMC_MULTI(ctx, true, rrrmc_snapshot_get(c, slot, at_row(chunks, r0 * nch_of(ctx))));
    if (!ctx) return RRRMC_ERR_INVALID_ARG;
    if (!chunks) return fail(ctx, RRRMC_ERR_INVALID_ARG, "chunks is NULL");
    if (slot < 0 || slot >= ctx->snap_slots || !ctx->snap_valid[(size_t)slot])
        return fail(ctx, RRRMC_ERR_INVALID_ARG, "snapshot slot %d is out of range or empty", slot);
    return spins_to_chunks(ctx, ctx->snap + native_spin_bytes(ctx) * (size_t)slot, chunks);
}

int32_t rrrmc_overlaps(rrrmc_ctx* ctx, int64_t npairs, const int32_t* slotA, const int32_t* slotB, int32_t* q_out)
{
    if (is_multi(ctx)) {
        if (npairs <= 0 || !q_out) return npairs == 0 ? RRRMC_OK : fail(ctx, RRRMC_ERR_INVALID_ARG, "npairs must be >= 0 and q_out not NULL");
        std::vector<std::vector<int32_t>> q(ctx->kids.size());
        for (size_t d = 0; d < q.size(); ++d) q[d].resize((size_t)(npairs * ctx->kids[d]->R));
        const int32_t rc = multi_each(ctx, [&](rrrmc_ctx* c, int64_t r0, int64_t) -> int32_t {
            size_t d = 0;
            while (ctx->kid_r0[d] != r0) ++d;
            return rrrmc_overlaps(c, npairs, slotA, slotB, q[d].data());
        }, true);
        if (rc) return rc;
        for (size_t d = 0; d < q.size(); ++d)
            for (int64_t pp = 0; pp < npairs; ++pp)
                std::memcpy(q_out + pp * ctx->R + ctx->kid_r0[d], q[d].data() + pp * ctx->kids[d]->R, sizeof(int32_t) * (size_t)ctx->kids[d]->R);
        return RRRMC_OK;
    }
    int32_t rc = ensure_state(ctx, true);
    if (rc) return rc;
    if (npairs < 0) return fail(ctx, RRRMC_ERR_INVALID_ARG, "npairs must be >= 0");
    if (npairs == 0) return RRRMC_OK;
    if (!slotA || !slotB || !q_out) return fail(ctx, RRRMC_ERR_INVALID_ARG, "slotA, slotB, q_out must not be NULL");
    if (npairs > 65535) return fail(ctx, RRRMC_ERR_INVALID_ARG, "at most 65535 pairs per call, given %lld", (long long)npairs);
    const size_t sb = native_spin_bytes(ctx);
    std::vector<const void*> tab((size_t)(2 * npairs));
    for (int64_t p = 0; p < npairs; ++p)
        for (int side = 0; side < 2; ++side) {
            const int32_t sl = side ? slotB[p] : slotA[p];
            if (sl == -1) { tab[(size_t)(side * npairs + p)] = native_spins(ctx); continue; }     // the live configuration
            if (sl < 0 || sl >= ctx->snap_slots || !ctx->snap_valid[(size_t)sl])
                return fail(ctx, RRRMC_ERR_INVALID_ARG, "pair %lld: snapshot slot %d is out of range or empty", (long long)p, sl);
            tab[(size_t)(side * npairs + p)] = ctx->snap + sb * (size_t)sl;
        }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if ((size_t)npairs > ctx->pairs_cap) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        free_dev(ctx->d_pairs); free_dev(ctx->d_ovl);
        ctx->pairs_cap = 0;
        HIP_TRY(ctx, hipMalloc(&ctx->d_pairs, sizeof(void*) * 2 * (size_t)npairs));
        HIP_TRY(ctx, hipMalloc(&ctx->d_ovl, sizeof(int32_t) * (size_t)npairs * (size_t)ctx->Rpad));
        ctx->pairs_cap = (size_t)npairs;
    }
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_pairs, tab.data(), sizeof(void*) * tab.size(), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));      // tab is pageable host memory
    void** pa = ctx->d_pairs;
    void** pb = ctx->d_pairs + npairs;
    if (ctx->model == RRRMC_MODEL_SPARSE_F64) {
        hipLaunchKernelGGL(overlap_lanes_kernel, dim3((unsigned)ctx->pfW, (unsigned)npairs), dim3(64), 0, ctx->stream,
                           (const unsigned long long* const*)pa, (const unsigned long long* const*)pb, (int)ctx->N, (int)ctx->Rpad, ctx->d_ovl);
    } else if (chunk_layout(ctx)) {
        hipLaunchKernelGGL(overlap_chunks_kernel, dim3((unsigned)ctx->R, (unsigned)npairs), dim3(64), 0, ctx->stream,
                           (const uint32_t* const*)pa, (const uint32_t* const*)pb, (int)ctx->N, (int)ctx->qW, (int)ctx->Rpad, ctx->d_ovl);
    } else if (ctx->model == RRRMC_MODEL_SK_NORMAL || ctx->model == RRRMC_MODEL_SK_BINARY) {
        hipLaunchKernelGGL(overlap_b8_kernel, dim3((unsigned)ctx->G8, (unsigned)npairs), dim3(256), 0, ctx->stream,
                           (const uint8_t* const*)pa, (const uint8_t* const*)pb, (int)ctx->N, (int)ctx->Rpad, ctx->d_ovl);
    } else {
        hipLaunchKernelGGL(overlap_bs32_kernel, dim3((unsigned)ctx->G, (unsigned)npairs), dim3(256), 0, ctx->stream,
                           (const uint32_t* const*)pa, (const uint32_t* const*)pb, (int)ctx->N, (int)ctx->Rpad, ctx->d_ovl);
    }
    HIP_TRY(ctx, hipGetLastError());
    std::vector<int32_t> h((size_t)npairs * (size_t)ctx->Rpad);
    HIP_TRY(ctx, hipMemcpyAsync(h.data(), ctx->d_ovl, sizeof(int32_t) * h.size(), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (int64_t p = 0; p < npairs; ++p)
        std::memcpy(q_out + p * ctx->R, h.data() + p * ctx->Rpad, sizeof(int32_t) * (size_t)ctx->R);
    return RRRMC_OK;
}

int32_t rrrmc_quant_observables(rrrmc_ctx* ctx, double beta, double Gamma, double* Qenergy_out, double* tmag_out, double* ovs_out)
{
    RRRMC_MULTI(ctx, true, rrrmc_quant_observables(c, beta, Gamma, at_row(Qenergy_out, r0), at_row(tmag_out, r0), at_row(ovs_out, r0 * (c->qM / 2))));
    int32_t rc = ensure_state(ctx, true);
    if (rc) return rc;
    if (ctx->model != RRRMC_MODEL_QUANT_RRG) return fail(ctx, RRRMC_ERR_STATE, "rrrmc_quant_observables needs a GraphQuant context");
    if (!(ctx->last_fourK > 0.0)) return fail(ctx, RRRMC_ERR_STATE, "observables of a GraphQuant need fourK: call rrrmc_quant_set_field first");
    if (ctx->q_skn || ctx->q_spf) return fail(ctx, RRRMC_ERR_UNSUPPORTED, "Qenergy / overlaps are not wired for GraphSKNormal / Float64 sparse slices");
    const int64_t R = ctx->R, M = ctx->qM, Nk = ctx->qNk, N = ctx->N, H = M / 2;
    const size_t lds = sizeof(uint32_t) * (size_t)(M * ((Nk + 31) / 32) + M + H + 1);
    if (lds > (size_t)64 * 1024) return fail(ctx, RRRMC_ERR_UNSUPPORTED, "N = %lld spins per replica do not fit the observables kernel's LDS", (long long)N);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!ctx->d_qobs) HIP_TRY(ctx, hipMalloc(&ctx->d_qobs, sizeof(int32_t) * (size_t)R * (size_t)(1 + M + H)));
    int32_t* d_e0 = ctx->d_qobs;
    int32_t* d_es = d_e0 + R;
    int32_t* d_ov = d_es + R * M;
    hipLaunchKernelGGL(quant_observables_kernel, dim3((unsigned)R), dim3(256), lds, ctx->stream, ctx->q_spins, ctx->d_A, ctx->d_J,
                       ctx->q_sk ? ctx->q_Jb : nullptr, (int)ctx->q_Wk, (int)Nk, (int)M, (int)ctx->K, (int)ctx->qW, d_e0, d_es, d_ov);
    HIP_TRY(ctx, hipGetLastError());
    std::vector<int32_t> h((size_t)R * (size_t)(1 + M + H));
    HIP_TRY(ctx, hipMemcpyAsync(h.data(), ctx->d_qobs, sizeof(int32_t) * h.size(), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    // Float64 tail in the reference's operation order (QT.jl:113-122, 235-251, 253-268); beta here is the caller's, as in
    // transverse_mag(X0, C, beta), fourK the context's
    const double x = beta * ctx->last_fourK / 2;
    const double ch = std::cosh(x), sh = std::sinh(x);
    for (int64_t r = 0; r < R; ++r) {
        const double p = -(double)h[(size_t)r] / (double)N;
        const double tm = ch - p * sh;
        double E = -Gamma * tm;
        const double sN = std::sqrt((double)Nk);                  // GraphSK slices: energy(X1[k], C1[k]) = n / sqrt(Nk), SK.jl:49,95
        for (int64_t k = 0; k < M; ++k) {
            const double Ek = ctx->q_sk ? (double)h[(size_t)(R + r * M + k)] / sN : (double)h[(size_t)(R + r * M + k)];
            E += Ek / (double)N;
        }
        if (tmag_out) tmag_out[r] = tm;
        if (Qenergy_out) Qenergy_out[r] = E;
        if (ovs_out) {
            for (int64_t d = 1; d <= H; ++d) {
                double o = (double)h[(size_t)(R + R * M + r * H + d - 1)];
                if (d <= (M - 1) / 2) o /= (double)(M * Nk);
                else o /= (double)(M * Nk) / 2;          // even M, d = M/2: each slice has one partner at that distance
                ovs_out[r * H + d - 1] = o;
            }
        }
    }
    return RRRMC_OK;
}

namespace {
// Shared checks of the level-unit sparse graphs (GraphRRG / GraphEA ctor: RRG.jl:122-139, EA.jl:145-169): table ranges, sorted rows,
// couplings among the levels, bond symmetry (rJ, when given, too); computes allΔE(X) (RRG.jl:268-281, EA.jl:295-309) into dElist.
int32_t validate_level_graph(rrrmc_ctx* ctx, const int32_t* A, const int8_t* dJ, const double* rJ, const int32_t* lev, int32_t nlev,
                             int32_t ea_form, int* dElist, int* L_out, int cap)
{
    if (nlev < 1 || nlev > 16) return fail(ctx, RRRMC_ERR_INVALID_ARG, "between 1 and 16 levels are supported, given %d", nlev);
    for (int32_t a = 0; a < nlev; ++a) {
        if (lev[a] < -127 || lev[a] > 127) return fail(ctx, RRRMC_ERR_UNSUPPORTED, "level %d does not fit the int8 coupling table", lev[a]);
        for (int32_t b = 0; b < a; ++b)
            if (lev[a] == lev[b]) return fail(ctx, RRRMC_ERR_INVALID_ARG, "repeated levels in LEV");     // RRG.jl:100
    }
    const int64_t N = ctx->N, K = ctx->K;
    for (int64_t q = 0; q < N * K; ++q) {
        if (A[q] < 0 || A[q] >= N) return fail(ctx, RRRMC_ERR_INVALID_ARG, "A[%lld] = %d out of range 0..%lld", (long long)q, A[q], (long long)(N - 1));
        if (A[q] == q / K) return fail(ctx, RRRMC_ERR_INVALID_ARG, "self loop at site %lld", (long long)(q / K));
        if (q % K && A[q] < A[q - 1]) return fail(ctx, RRRMC_ERR_INVALID_ARG, "row %lld of A is not sorted", (long long)(q / K));
        if (rJ && !std::isfinite(rJ[q])) return fail(ctx, RRRMC_ERR_INVALID_ARG, "rJ[%lld] is not finite", (long long)q);
        bool ok = false;
        for (int32_t a = 0; a < nlev; ++a) ok |= (int32_t)dJ[q] == lev[a];
        if (!ok) return fail(ctx, RRRMC_ERR_INVALID_ARG, "the given J is incompatible with the levels (J[%lld] = %d)", (long long)q, (int)dJ[q]);   // RRG.jl:130
        if (!ea_form && q % K && A[q] == A[q - 1]) return fail(ctx, RRRMC_ERR_INVALID_ARG, "repeated neighbour in row %lld: pass ea_form = 1 for GraphEA tables", (long long)(q / K));
    }
    std::vector<uint8_t> used((size_t)(N * K), 0);
    for (int64_t x = 0; x < N; ++x)
        for (int64_t k = 0; k < K; ++k) {
            const int64_t y = A[x * K + k];
            bool found = false;
            for (int64_t l = 0; l < K && !found; ++l)
                if (!used[y * K + l] && A[y * K + l] == x && dJ[y * K + l] == dJ[x * K + k] && (!rJ || rJ[y * K + l] == rJ[x * K + k])) { used[y * K + l] = 1; found = true; }
            if (!found) return fail(ctx, RRRMC_ERR_INVALID_ARG, "bond (%lld,%lld) is not symmetric in (A, J)", (long long)x, (long long)y);
        }
    // allΔE(X0): sums of K terms +-l (RRG.jl:268-281, EA.jl:295-309)
    int64_t amax = 0;
    for (int32_t a = 0; a < nlev; ++a) amax = std::max<int64_t>(amax, lev[a] < 0 ? -(int64_t)lev[a] : lev[a]);
    const int64_t span = K * amax;
    std::vector<uint8_t> cur((size_t)(2 * span + 1), 0), nxt((size_t)(2 * span + 1), 0);
    cur[(size_t)span] = 1;
    for (int64_t n = 0; n < K; ++n) {
        std::fill(nxt.begin(), nxt.end(), 0);
        for (int64_t v = -span; v <= span; ++v)
            if (cur[(size_t)(v + span)])
                for (int32_t a = 0; a < nlev; ++a) {
                    if (v + lev[a] >= -span && v + lev[a] <= span) nxt[(size_t)(v + lev[a] + span)] = 1;
                    if (v - lev[a] >= -span && v - lev[a] <= span) nxt[(size_t)(v - lev[a] + span)] = 1;
                }
        cur.swap(nxt);
    }
    int L = 0;
    for (int64_t a = 0; a <= span; ++a)
        if (cur[(size_t)(span + a)] || cur[(size_t)(span - a)]) {
            if (L == cap) return fail(ctx, RRRMC_ERR_UNSUPPORTED, "more than %d energy levels: not covered by the integer-level kernels", cap);
            dElist[L++] = (int)(2 * a);
        }
    *L_out = L;
    return RRRMC_OK;
}
}  // namespace

int32_t rrrmc_set_graph_discretized(rrrmc_ctx* ctx, const int32_t* A, const int8_t* dJ, const double* rJ, const int32_t* lev, int32_t nlev,
                                    int32_t ea_form)
{
    RRRMC_MULTI(ctx, false, rrrmc_set_graph_discretized(c, A, dJ, rJ, lev, nlev, ea_form));
    smp_drop(ctx);
    if (!ctx) return RRRMC_ERR_INVALID_ARG;
    if (ctx->model != RRRMC_MODEL_SPARSE_DISCRETIZED) return fail(ctx, RRRMC_ERR_STATE, "rrrmc_set_graph_discretized is for RRRMC_MODEL_SPARSE_DISCRETIZED");
    if (!A || !dJ || !rJ || !lev) return fail(ctx, RRRMC_ERR_INVALID_ARG, "A, dJ, rJ, lev must not be NULL");
    int L = 0;
    int32_t rc = validate_level_graph(ctx, A, dJ, rJ, lev, nlev, ea_form, ctx->db_dElist, &L, kDLmax);
    if (rc) return rc;
    const int64_t N = ctx->N, K = ctx->K;
    ctx->db_L = L;
    ctx->db_ea_form = ea_form ? 1 : 0;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(ctx->d_A, A, sizeof(int32_t) * N * K, hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(ctx->db_dJ, dJ, sizeof(int8_t) * N * K, hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(ctx->db_rJ, rJ, sizeof(double) * N * K, hipMemcpyHostToDevice));
    ctx->graph_set = true;
    return RRRMC_OK;
}

int32_t rrrmc_set_graph_levels(rrrmc_ctx* ctx, const int32_t* A, const int8_t* J, const int32_t* lev, int32_t nlev, int32_t ea_form)
{
    RRRMC_MULTI(ctx, false, rrrmc_set_graph_levels(c, A, J, lev, nlev, ea_form));
    smp_drop(ctx);
    if (!ctx) return RRRMC_ERR_INVALID_ARG;
    if (ctx->model != RRRMC_MODEL_SPARSE_LEVELS) return fail(ctx, RRRMC_ERR_STATE, "rrrmc_set_graph_levels is for RRRMC_MODEL_SPARSE_LEVELS");
    if (!A || !J || !lev) return fail(ctx, RRRMC_ERR_INVALID_ARG, "A, J, lev must not be NULL");
    LevTable lv{};
    int32_t rc = validate_level_graph(ctx, A, J, nullptr, lev, nlev, ea_form, lv.dElist, &lv.L, kSLmax);
    if (rc) return rc;
    lv.skip_zero = ea_form ? 0 : 1;          // neighbors(X, i): non-zero couplings for GraphRRG (RRG.jl:133), de-duplicated A[i] for GraphEA (EA.jl:158)
    const int64_t N = ctx->N, K = ctx->K;
    ctx->lv = lv;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(ctx->d_A, A, sizeof(int32_t) * N * K, hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(ctx->d_J, J, sizeof(int8_t) * N * K, hipMemcpyHostToDevice));
    // rrrMC / extremal_opt class arrays depend on L: drop those of a previous graph
    free_dev(ctx->rp_cls); free_dev(ctx->rp_sv); free_dev(ctx->rp_spos);
    ctx->graph_set = true;
    return RRRMC_OK;
}

int32_t rrrmc_set_level_scale(rrrmc_ctx* ctx, int64_t mul, double div)
{
    RRRMC_MULTI(ctx, false, rrrmc_set_level_scale(c, mul, div));
    smp_drop(ctx);
    if (!ctx) return RRRMC_ERR_INVALID_ARG;
    if (ctx->model != RRRMC_MODEL_SPARSE_DISCRETIZED && ctx->model != RRRMC_MODEL_SPARSE_LEVELS)
        return fail(ctx, RRRMC_ERR_STATE, "rrrmc_set_level_scale is for RRRMC_MODEL_SPARSE_DISCRETIZED and RRRMC_MODEL_SPARSE_LEVELS");
    if (mul < 1 || !(div > 0.0) || !std::isfinite(div)) return fail(ctx, RRRMC_ERR_INVALID_ARG, "level scale must have mul >= 1 and a finite div > 0");
    // the largest level sum the kernels form is N * K * 127 units (the energy): keep units * mul inside the Float64-exact integers
    if ((double)mul * 127.0 * (double)ctx->N * (double)ctx->K >= 9007199254740992.0)
        return fail(ctx, RRRMC_ERR_UNSUPPORTED, "level scale mul=%lld too large for N=%lld, K=%lld", (long long)mul, (long long)ctx->N, (long long)ctx->K);
    ctx->db_lev_mul = mul; ctx->db_lev_div = div;
    ctx->lv_mul = mul; ctx->lv_div = div;
    return RRRMC_OK;
}

int32_t rrrmc_discretize_scaled(const double* x, int64_t n, const int32_t* lev, int32_t nlev, int64_t mul, double div, int8_t* d_out, double* r_out)
{
    // discretize: src/Common.jl:38-49 (nearest level, the first one on ties; residual = x - level); the level's Float64 value is
    // (units * mul) / div — DFloat64 levels promote to Float64 in `x - d` (src/DFloats.jl:26,42)
    if (!x || !lev || !d_out || !r_out || nlev < 1 || n < 0 || mul < 1 || !(div > 0.0)) return RRRMC_ERR_INVALID_ARG;
    for (int64_t q = 0; q < n; ++q) {
        int32_t d = lev[0];
        double r = x[q] - (double)((int64_t)d * mul) / div;
        for (int32_t l = 1; l < nlev; ++l) {
            const double r1 = x[q] - (double)((int64_t)lev[l] * mul) / div;
            if (std::fabs(r1) < std::fabs(r)) { d = lev[l]; r = r1; }
        }
        if (d < -127 || d > 127) return RRRMC_ERR_UNSUPPORTED;
        d_out[q] = (int8_t)d;
        r_out[q] = r;
    }
    return RRRMC_OK;
}

int32_t rrrmc_discretize(const double* x, int64_t n, const int32_t* lev, int32_t nlev, int8_t* d_out, double* r_out)
{
    return rrrmc_discretize_scaled(x, n, lev, nlev, 1, 1.0, d_out, r_out);
}

namespace {
// n-th normal of the GAUSS stream: Box-Muller on the two 53-bit uniforms of Philox block n >> 1 (cos / sin branch)  [randn()]
double gauss_draw(uint64_t seed, uint64_t n)
{
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    const uint64_t blk = n >> 1;
    const Philox4 o = philox4x32_10((uint32_t)blk, (uint32_t)(blk >> 32), 0u, TAG_GAUSS, k0, k1);
    const uint64_t a = ((uint64_t)o.w[0] << 32) | o.w[1], b = ((uint64_t)o.w[2] << 32) | o.w[3];
    const double u1 = ((double)(a >> 11) + 1.0) * 0x1.0p-53, u2 = (double)(b >> 11) * 0x1.0p-53;
    const double rad = std::sqrt(-2.0 * std::log(u1)), ang = 6.283185307179586476925286766559 * u2;
    return (n & 1u) ? rad * std::sin(ang) : rad * std::cos(ang);
}
}  // namespace

int32_t rrrmc_set_graph_f64(rrrmc_ctx* ctx, const int32_t* A, const double* J)
{
    RRRMC_MULTI(ctx, false, rrrmc_set_graph_f64(c, A, J));
    smp_drop(ctx);
    if (!ctx) return RRRMC_ERR_INVALID_ARG;
    const bool qspf = ctx->model == RRRMC_MODEL_QUANT_RRG && ctx->q_spf;          // the slice graph of a GraphQEAT: (A, J) is Nk x K
    if (ctx->model != RRRMC_MODEL_SPARSE_F64 && !qspf) return fail(ctx, RRRMC_ERR_STATE, "rrrmc_set_graph_f64 is for RRRMC_MODEL_SPARSE_F64 (or a context made by rrrmc_ctx_create_quant_f64)");
    if (!A || !J) return fail(ctx, RRRMC_ERR_INVALID_ARG, "A and J must not be NULL");
    const int64_t N = qspf ? ctx->qNk : ctx->N, K = ctx->K;
    for (int64_t q = 0; q < N * K; ++q) {
        if (A[q] < 0 || A[q] >= N) return fail(ctx, RRRMC_ERR_INVALID_ARG, "A[%lld] = %d out of range 0..%lld", (long long)q, A[q], (long long)(N - 1));
        if (!std::isfinite(J[q])) return fail(ctx, RRRMC_ERR_INVALID_ARG, "J[%lld] is not finite", (long long)q);
        if (A[q] == q / K) return fail(ctx, RRRMC_ERR_INVALID_ARG, "self loop at site %lld", (long long)(q / K));
        if (q % K && A[q] < A[q - 1]) return fail(ctx, RRRMC_ERR_INVALID_ARG, "row %lld of A is not sorted", (long long)(q / K));   // EA.jl:46, RRG.jl:64
    }
    // every bond must appear from both ends with the same coupling (multi-edges allowed: GraphEANormal with L = 2)
    std::vector<uint8_t> used((size_t)(N * K), 0);
    for (int64_t x = 0; x < N; ++x)
        for (int64_t k = 0; k < K; ++k) {
            const int64_t y = A[x * K + k];
            bool found = false;
            for (int64_t l = 0; l < K && !found; ++l)
                if (!used[y * K + l] && A[y * K + l] == x && J[y * K + l] == J[x * K + k]) { used[y * K + l] = 1; found = true; }
            if (!found) return fail(ctx, RRRMC_ERR_INVALID_ARG, "bond (%lld,%lld) is not symmetric in (A, J)", (long long)x, (long long)y);
        }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(ctx->d_A, A, sizeof(int32_t) * N * K, hipMemcpyHostToDevice));
    if (qspf) {
        HIP_TRY(ctx, hipMemcpy(ctx->q_Jf, J, sizeof(double) * N * K, hipMemcpyHostToDevice));
        ctx->q_cache_valid = false;
        ctx->graph_set = true;
        return RRRMC_OK;
    }
    HIP_TRY(ctx, hipMemcpy(ctx->pf_J, J, sizeof(double) * N * K, hipMemcpyHostToDevice));
    ctx->h_A.assign(A, A + N * K);
    ctx->h_Jf.assign(J, J + N * K);
    ctx->pf_multi_edge = false;
    for (int64_t x = 0; x < N; ++x)
        for (int64_t a = 1; a < K; ++a)
            for (int64_t b = 0; b < a; ++b)
                if (A[x * K + a] == A[x * K + b]) ctx->pf_multi_edge = true;
    if (ctx->pff_ready) {            // a new graph: the fast mode's tables are rebuilt on its next call
        free_dev(ctx->pff_table); free_dev(ctx->pff_absJ); free_dev(ctx->pff_bond_off); free_dev(ctx->pff_thr_hi); free_dev(ctx->pff_thr_lo); free_dev(ctx->pff_flags);
        ctx->pff_ready = false;
    }
    ctx->graph_set = true;
    return RRRMC_OK;
}

int32_t rrrmc_gen_couplings_gauss(int64_t N, int64_t K, const int32_t* A, uint64_t seed, double* J_out)
{
    if (N < 1 || K < 1 || !A || !J_out) return RRRMC_ERR_INVALID_ARG;
    std::vector<uint8_t> filled((size_t)(N * K), 0);
    uint64_t ndraw = 0;
    for (int64_t x = 0; x < N; ++x)
        for (int64_t k = 0; k < K; ++k) {
            const int64_t y = A[x * K + k];
            if (y < 0 || y >= N) return RRRMC_ERR_INVALID_ARG;
            if (x < y) {
                const double Jxy = gauss_draw(seed, ndraw++);
                if (filled[x * K + k]) return RRRMC_ERR_INVALID_ARG;
                J_out[x * K + k] = Jxy; filled[x * K + k] = 1;
                int64_t l = 0;
                while (l < K && filled[y * K + l]) ++l;          // findfirst(==(m), J[y]): RRG.jl:84
                if (l == K) return RRRMC_ERR_INVALID_ARG;
                J_out[y * K + l] = Jxy; filled[y * K + l] = 1;
            }
        }
    for (int64_t q = 0; q < N * K; ++q) if (!filled[q]) return RRRMC_ERR_INVALID_ARG;
    return RRRMC_OK;
}

int32_t rrrmc_gen_sk_binary(int64_t N, uint64_t seed, uint64_t* Jc)
{
    // gen_J, src/graphs/SK.jl:17-26: rows of bitrand(N), zero diagonal, upper triangle mirrored.  SKBITS stream:
    // bit j of row i = bit (j & 31) of word ((j >> 5) & 3) of ctr (j >> 7, i, 0, TAG_SKBITS)
    if (!Jc) return fail(nullptr, RRRMC_ERR_INVALID_ARG, "J_chunks_out is NULL");
    if (N < 1) return fail(nullptr, RRRMC_ERR_INVALID_ARG, "N must be >= 1");
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    const int64_t nch = (N + 63) / 64;
    std::memset(Jc, 0, sizeof(uint64_t) * N * nch);
    for (int64_t i = 0; i < N; ++i)
        for (int64_t jb = 0; jb < N; jb += 128) {
            const Philox4 o = philox4x32_10((uint32_t)(jb >> 7), (uint32_t)i, 0u, TAG_SKBITS, k0, k1);
            for (int64_t j = jb; j < N && j < jb + 128; ++j)
                if ((o.w[(j >> 5) & 3] >> (j & 31)) & 1u) Jc[i * nch + (j >> 6)] |= 1ull << (j & 63);
        }
    for (int64_t i = 0; i < N; ++i) {
        Jc[i * nch + (i >> 6)] &= ~(1ull << (i & 63));
        for (int64_t j = i + 1; j < N; ++j) {
            const uint64_t b = (Jc[i * nch + (j >> 6)] >> (j & 63)) & 1ull;
            Jc[j * nch + (i >> 6)] = (Jc[j * nch + (i >> 6)] & ~(1ull << (i & 63))) | (b << (i & 63));
        }
    }
    return RRRMC_OK;
}

int32_t rrrmc_gen_sk_gauss(int64_t N, uint64_t seed, double* J_out)
{
    // gen_J_gauss, src/graphs/SK.jl:170-179: rows of randn(N) scaled by 1/sqrt(N), zero diagonal, upper triangle mirrored.
    if (!J_out) return fail(nullptr, RRRMC_ERR_INVALID_ARG, "J_out is NULL");
    if (N < 1) return fail(nullptr, RRRMC_ERR_INVALID_ARG, "N must be >= 1");
    const double scale = 1.0 / std::sqrt((double)N);
    for (int64_t i = 0; i < N; ++i)
        for (int64_t j = 0; j < N; ++j) J_out[i * N + j] = gauss_draw(seed, (uint64_t)(i * N + j)) * scale;
    for (int64_t i = 0; i < N; ++i) {
        J_out[i * N + i] = 0.0;
        for (int64_t j = i + 1; j < N; ++j) J_out[j * N + i] = J_out[i * N + j];
    }
    return RRRMC_OK;
}

// ---- host-side graph constructors -------------------------------------------------------------------

int32_t rrrmc_gen_rrg(int64_t N, int64_t K, uint64_t seed, int32_t* A_out)
{
    // gen_RRG, src/graphs/RRG.jl:26-69: pairing model, restart of the whole attempt on a self loop or a double edge
    if (!A_out) return fail(nullptr, RRRMC_ERR_INVALID_ARG, "A_out is NULL");
    if (K < 1 || N < 1) return fail(nullptr, RRRMC_ERR_INVALID_ARG, "N and K must be >= 1");
    if ((N * K) % 2) return fail(nullptr, RRRMC_ERR_INVALID_ARG, "N * K must be even, given N=%lld, K=%lld", (long long)N, (long long)K);
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    const int64_t NK = N * K;
    std::vector<int64_t> stubs((size_t)NK);
    std::vector<std::vector<int32_t>> nb((size_t)N);
    uint64_t ndraw = 0;
    for (int attempt = 0; attempt < 100000; ++attempt) {
        for (int64_t q = 0; q < NK; ++q) stubs[q] = q;
        for (auto& v : nb) v.clear();
        int64_t len = NK;
        bool bad = false;
        while (len > 0 && !bad) {
            const int64_t j = (int64_t)mulhi64(stream_u64(k0, k1, TAG_GRAPH, ndraw++), (uint64_t)(len - 1));   // rand(1:(l-1)) - 1
            const int64_t v1 = stubs[len - 1] % N;
            std::swap(stubs[j], stubs[len - 2]);
            const int64_t v2 = stubs[len - 2] % N;
            len -= 2;
            bad = (v1 == v2);
            for (int32_t y : nb[v1]) bad = bad || (y == v2);
            if (!bad) { nb[v1].push_back((int32_t)v2); nb[v2].push_back((int32_t)v1); }
        }
        if (bad) continue;
        for (int64_t x = 0; x < N; ++x) {
            std::vector<int32_t>& v = nb[x];
            for (size_t a = 1; a < v.size(); ++a) for (size_t b2 = a; b2 > 0 && v[b2 - 1] > v[b2]; --b2) std::swap(v[b2 - 1], v[b2]);
            for (int64_t k = 0; k < K; ++k) A_out[x * K + k] = v[k];
        }
        return RRRMC_OK;
    }
    return fail(nullptr, RRRMC_ERR_UNSUPPORTED, "gen_rrg: no simple graph after 100000 attempts (K too large?)");
}

int32_t rrrmc_gen_ea(int64_t L, int64_t D, int32_t* A_out)
{
    // gen_EA, src/graphs/EA.jl:24-43
    if (!A_out) return fail(nullptr, RRRMC_ERR_INVALID_ARG, "A_out is NULL");
    if (L < 2) return fail(nullptr, RRRMC_ERR_INVALID_ARG, "L must be >= 2, given: %lld", (long long)L);
    if (D < 1) return fail(nullptr, RRRMC_ERR_INVALID_ARG, "D must be >= 1, given: %lld", (long long)D);
    int64_t N = 1;
    for (int64_t d = 0; d < D; ++d) N *= L;
    std::vector<std::vector<int32_t>> nb((size_t)N);
    for (int64_t x = 0; x < N; ++x) {
        int64_t stride = 1;
        for (int64_t d = 0; d < D; ++d, stride *= L) {
            const int64_t c = (x / stride) % L;
            const int64_t y = x + (((c + 1) % L) - c) * stride;
            nb[x].push_back((int32_t)y);
            nb[y].push_back((int32_t)x);
        }
    }
    for (int64_t x = 0; x < N; ++x) {
        std::vector<int32_t>& v = nb[x];
        for (size_t a = 1; a < v.size(); ++a) for (size_t b = a; b > 0 && v[b - 1] > v[b]; --b) std::swap(v[b - 1], v[b]);
        for (int64_t k = 0; k < 2 * D; ++k) A_out[x * 2 * D + k] = v[k];
    }
    return RRRMC_OK;
}

int32_t rrrmc_gen_couplings_pm1(int64_t N, int64_t K, const int32_t* A, uint64_t seed, int8_t* J_out)
{
    // gen_J, src/graphs/RRG.jl:71-96 / src/graphs/EA.jl:45-71 with rand((-1, 1)) (RRG.jl:154-156)
    if (!A || !J_out) return fail(nullptr, RRRMC_ERR_INVALID_ARG, "A / J_out is NULL");
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    std::vector<int64_t> filled((size_t)N, 0);   // next unfilled slot of J[y]
    std::memset(J_out, 0, (size_t)(N * K));
    uint64_t ndraw = 0;
    for (int64_t x = 0; x < N; ++x)
        for (int64_t k = 0; k < K; ++k) {
            const int64_t y = A[x * K + k];
            if (y < 0 || y >= N) return fail(nullptr, RRRMC_ERR_INVALID_ARG, "A[%lld] out of range", (long long)(x * K + k));
            if (x < y) {
                const int8_t Jxy = mulhi64(stream_u64(k0, k1, TAG_COUPLING, ndraw++), 2) ? 1 : -1;
                if (J_out[x * K + k] != 0) return fail(nullptr, RRRMC_ERR_INVALID_ARG, "inconsistent neighbour table at site %lld", (long long)x);
                J_out[x * K + k] = Jxy;
                while (filled[y] < K && J_out[y * K + filled[y]] != 0) filled[y]++;
                if (filled[y] >= K) return fail(nullptr, RRRMC_ERR_INVALID_ARG, "inconsistent neighbour table at site %lld", (long long)y);
                J_out[y * K + filled[y]] = Jxy;
            }
        }
    for (int64_t q = 0; q < N * K; ++q)
        if (J_out[q] == 0) return fail(nullptr, RRRMC_ERR_INVALID_ARG, "neighbour table is not symmetric");
    return RRRMC_OK;
}

int32_t rrrmc_gen_couplings_lev(int64_t N, int64_t K, const int32_t* A, uint64_t seed, const int32_t* lev, int32_t nlev, int8_t* J_out)
{
    // gen_J, src/graphs/RRG.jl:71-96 / src/graphs/EA.jl:45-71 with rand(vLEV) (RRG.jl:154-156): COUPLING stream, one draw per bond
    if (!A || !J_out || !lev || nlev < 1) return fail(nullptr, RRRMC_ERR_INVALID_ARG, "A / J_out / lev is NULL or nlev < 1");
    for (int32_t a = 0; a < nlev; ++a)
        if (lev[a] < -127 || lev[a] > 127) return fail(nullptr, RRRMC_ERR_UNSUPPORTED, "level %d does not fit the int8 coupling table", lev[a]);
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    std::vector<uint8_t> set((size_t)(N * K), 0);
    std::memset(J_out, 0, (size_t)(N * K));
    uint64_t ndraw = 0;
    for (int64_t x = 0; x < N; ++x)
        for (int64_t k = 0; k < K; ++k) {
            const int64_t y = A[x * K + k];
            if (y < 0 || y >= N) return fail(nullptr, RRRMC_ERR_INVALID_ARG, "A[%lld] out of range", (long long)(x * K + k));
            if (x < y) {
                const int8_t Jxy = (int8_t)lev[mulhi64(stream_u64(k0, k1, TAG_COUPLING, ndraw++), (uint64_t)nlev)];
                if (set[(size_t)(x * K + k)]) return fail(nullptr, RRRMC_ERR_INVALID_ARG, "inconsistent neighbour table at site %lld", (long long)x);
                J_out[x * K + k] = Jxy; set[(size_t)(x * K + k)] = 1;
                int64_t l = 0;
                while (l < K && set[(size_t)(y * K + l)]) ++l;
                if (l == K) return fail(nullptr, RRRMC_ERR_INVALID_ARG, "inconsistent neighbour table at site %lld", (long long)y);
                J_out[y * K + l] = Jxy; set[(size_t)(y * K + l)] = 1;
            }
        }
    for (int64_t q = 0; q < N * K; ++q)
        if (!set[(size_t)q]) return fail(nullptr, RRRMC_ERR_INVALID_ARG, "neighbour table is not symmetric");
    return RRRMC_OK;
}

}  // extern "C"
