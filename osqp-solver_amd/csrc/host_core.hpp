// host_core.hpp -- host side of the MI355X OSQP core: everything setup() does
// before the ADMM iterate moves to the GPU (SURVEY.md section 8(a) rows E1-E5,
// E13) plus the builder of the device schedules.  Pure C++17, no HIP types, so
// that CPU-only CI can exercise it through the mi_osqp_debug_* entry points.
//
// Nothing here is shared with oracle/: the oracle is an up-looking QDLDL-style
// restatement, this file is an independent left-looking factorisation with a
// supernode-aware, level-scheduled layout designed for wave64 execution.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

namespace miosqp {

constexpr double kInfty = 1e30;          // [REF] src/constraints/constraints.h:11
constexpr double kRhoMin = 1e-6, kRhoMax = 1e6, kRhoEqOverIneq = 1e3, kRhoTol = 1e-4;
constexpr double kMinScaling = 1e-4, kMaxScaling = 1e4;
constexpr double kDivisionTol = 1.0 / kInfty;
constexpr int kChunk = 16;               // rows of one in-block triangle (phase B)
constexpr uint32_t kNoRow = 0xFFFFFFFFu;

struct Settings {
  double rho = 0.1, sigma = 1e-6;
  int64_t scaling = 10, adaptive_rho = 1, adaptive_rho_interval = 0;
  double adaptive_rho_tolerance = 5.0;
  int64_t max_iter = 4000;
  double eps_abs = 1e-3, eps_rel = 1e-3, eps_prim_inf = 1e-4, eps_dual_inf = 1e-4, alpha = 1.6;
  int64_t scaled_termination = 0, check_termination = 25, warm_start = 1, verbose = 0;
};
int validate_settings(const Settings &s);   // 0 ok

// One pull-schedule: every target row t gets  xs[t] -= sum_k val[k] * xs[idx[k]].
//
// The schedule is a flat list of PHASES separated by workgroup barriers; in every
// phase each of the nw waves owns exactly one (possibly empty) contiguous range of
// wave-steps (64 lanes, one slot per lane).  Two kinds of phase:
//   A  row steps.  A step carries (lt, flush, out_base): groups of T = 2^lt lanes
//      accumulate one target row; on a flush step the groups are reduced and
//      applied to the 64/T rows listed at out_base.  Long rows span several steps
//      (flush on the last one), short rows are one flush step each.
//   B  the dense in-chunk triangle of a <=16-row chunk of a supernode, solved
//      column by column inside ONE wave (lane = (row i, QP b)); its 15 values per
//      lane are stored in the same step format (element k of lane (i,b) is
//      component k % BT of step k / BT), so one prefetch routine serves both kinds.
// A wave's steps of a phase are contiguous in memory: its values stream from HBM
// and are prefetched, unconditionally, one phase ahead.
//
// Logical slots (what `src`, `idx` index): A steps: step*64 + lane; B tasks:
// 64*n_steps + task*240 + k*16 + i.  Physical position of QP b's double inside a
// tile: A: slot*BT + b ; B: ((bstep0[task] + k/BT)*64 + i*BT + b)*BT + k%BT for k < BT * (steps of the task).
struct Schedule {
  int n_phases = 0, nw = 0, bt = 1, sb = 15;
  std::vector<uint32_t> phase;  // per phase, stride 4*nw+1: kind(0=A,1=B) then per wave (begin, end, out_base, 0)
  std::vector<int> level_first_phase;   // for replay/diagnostics: first phase of every level (+ end)
  std::vector<uint32_t> step;   // per A step: lt | flush << 3 | out_base << 4
  std::vector<uint32_t> outA;   // target rows of flush steps (64/T each), kNoRow = none
  std::vector<uint32_t> outB;   // kChunk rows per block task, processing order
  std::vector<uint32_t> idx;    // per A slot: gather index into the LDS vector
  std::vector<uint32_t> idxw;   // DEVICE index words, one per physical slot (phys_steps*64): low 16 bits = gather
                                // index, high 16 bits = target row of the lane's group on flush steps / of lane
                                // (i,b) in the first step of a block task (0xFFFF = none)
  std::vector<int32_t> src;     // per logical slot: canonical value index, -1 = structural zero
  uint32_t n_slots = 0;         // logical slots
  uint32_t n_steps = 0;         // A steps (incl. the all-zero padding step `zero_step`)
  uint32_t n_taskB = 0, zero_step = 0;
  std::vector<uint32_t> bstep0;  // per block task: first physical step (n_taskB + 1 entries); a task of r rows owns ceil((r-1)/bt) steps
  uint32_t phys_steps() const { return bstep0.empty() ? n_steps : bstep0.back(); }
  size_t phase_stride() const { return 4 * (size_t)nw + 1; }
  int n_levels = 0;
};

// Block form of the factor for the DEVICE refactorisation (row E13): L is cut
// into dense blocks B(I,J) = L[rows of chunk I, cols of chunk J] (column-major,
// h_I x w_J, zero padded), processed per chunk-column level:
//   U  B(I,J) -= sum_K B(I,K) D_K B(J,K)^T      (pull: one wave owns one target block)
//   D  LDL' of the diagonal block B(J,J)
//   T  B(I,J) <- B(I,J) (L_JJ D_J)^-T
struct BlockFactor {
  uint32_t storage = 0;               // doubles per QP of block storage
  std::vector<uint32_t> blk;          // 4/block: off, c0 of row chunk, c0 of col chunk, (h << 8) | w
  std::vector<uint32_t> lvl;          // 6/level: u_begin,u_end, d_begin,d_end, t_begin,t_end
  std::vector<uint32_t> utask;        // 4/task : block id, tri_begin, tri_mid, tri_end  ([begin,mid): rank-1 sources, i.e. 1-column chunks)
  std::vector<uint32_t> tri;          // 2/triple: block (I,K), block (J,K)
  std::vector<uint32_t> dtask;        // diagonal block ids
  std::vector<uint32_t> ttask;        // 2/task : block id, diagonal block id
  std::vector<uint32_t> asm_dst;      // per natural KKT entry: position in block storage
  std::vector<uint32_t> asm_src;      // per natural KKT entry: (kind << 29) | index
  std::vector<int32_t> lpos;          // canonical L entry -> position in block storage
  int n_levels = 0;
  size_t n_blocks() const { return blk.size() / 4; }
};
enum { ASM_P = 0, ASM_P_SIGMA = 1, ASM_SIGMA = 2, ASM_A = 3, ASM_NEG_RHOINV = 4 };

struct Analysis {
  int n = 0, m = 0, N = 0;
  // triu(P) and A patterns (CSC, 32-bit on our side)
  std::vector<int> Pp, Pi, Psrc;   // Psrc: index into the caller's P value array
  std::vector<int> Ap, Ai;
  // natural upper-triangular KKT [[P+sigma I, A'],[A, -1/rho]]
  std::vector<int> Kp, Ki, PtoK, AtoK, rhotoK, sigmaOnlyK;
  std::vector<char> PisDiag;
  // fill-reducing permutation (perm[new] = old)
  std::vector<int> perm, pinv;
  int ordering = 0;                // 0 minimum degree, 1 nested dissection (picked by analyze())
  // permuted KKT, lower triangle by columns, and map natural entry -> position
  std::vector<int> Klp, Kli, KtoKl;
  // symbolic factor: strictly-lower L by columns (sorted rows) + row view
  std::vector<int> Lp, Li, Rp, Rj, Rpos, etree;
  std::vector<int> sn_start;       // supernode boundaries
  std::vector<int> chunk_start;    // <=16-column chunks (phase-B blocks)
  std::vector<int> chunk_lev;      // forward level of every chunk
  Schedule fwd, bwd, chk;
  BlockFactor bf;
  std::vector<int32_t> fwd_srcblk, bwd_srcblk;   // fwd/bwd slot -> block-storage position (-1 = zero)
  int nnzL() const { return Lp.empty() ? 0 : Lp.back(); }
  int nnzK() const { return Kp.empty() ? 0 : Kp.back(); }
};

// E1 (pattern part), E4, E5-symbolic and the schedules.  Returns 0 or an error
// code of include/mi_osqp.h.
// `nwaves` = waves per workgroup the device kernels will run with, `bt` = QPs per
// tile (the step programs are laid out per wave; block tasks depend on bt).
int analyze(int64_t n, int64_t m, const int64_t *Pp, const int64_t *Pi, const int64_t *Ap,
            const int64_t *Ai, Analysis &an, int nwaves = 8, int bt = 1);
// physical position (in doubles, inside one tile) of QP b's value of a logical slot
size_t phys_index(const Schedule &s, uint32_t slot, int b);   // (size_t)-1: the slot has no storage (k beyond the task's steps)

// Per-QP numeric state kept on the host (needed for rescaling and refactors).
struct QPNumeric {
  std::vector<double> Pv, Av, q, l, u;            // scaled problem data
  std::vector<double> D, Dinv, E, Einv;           // Ruiz scaling
  double c = 1.0, cinv = 1.0;
  double rho = 0.1;
  std::vector<double> rho_vec, rho_inv;
  std::vector<int8_t> ctype;
  std::vector<double> Lx, Dl, Dlinv;              // factor (canonical CSC order)
};

void load_qp(const Analysis &an, const Settings &st, const double *Pval, const double *q,
             const double *Aval, const double *l, const double *u, QPNumeric &qp);
void scale_qp(const Analysis &an, const Settings &st, QPNumeric &qp);       // E2
void unscale_qp(const Analysis &an, QPNumeric &qp);
void set_rho_vec(const Analysis &an, const Settings &st, QPNumeric &qp);    // E3
// returns 1 if any constraint type changed (bounds update path of E13)
int refresh_rho_types(const Analysis &an, QPNumeric &qp);
void apply_rho(const Analysis &an, QPNumeric &qp, double rho_new);
// E5 numeric: left-looking LDL' on the permuted KKT. 0 ok, MI_OSQP_ERR_NONCONVEX else.
int factor_qp(const Analysis &an, const Settings &st, QPNumeric &qp, std::vector<double> &work);
// reference solve with the canonical factor (natural order in/out)
void direct_kkt_solve(const Analysis &an, const QPNumeric &qp, const double *rhs, double *sol);
// sequential interpreter of the device schedules (tests only; see mi_osqp.h)
void replay_kkt_solve(const Analysis &an, const QPNumeric &qp, const double *rhs, double *sol);
// host interpreter of the device block factorisation (tests only): fills Lx/Dlinv
// of `out` from qp's scaled data and rho vector exactly as factor_kernel does
int replay_block_factor(const Analysis &an, const Settings &st, const QPNumeric &qp, QPNumeric &out);
// combined value array the check-SpMV schedule indexes: [P triu | A]
void replay_spmv(const Analysis &an, const QPNumeric &qp, const double *x, const double *y,
                 double *Px, double *Aty, double *Ax);

}  // namespace miosqp
