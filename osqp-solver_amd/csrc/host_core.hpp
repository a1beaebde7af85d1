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

// One pull-schedule: every target row t gets  xs[t] -= sum_k val[k] * xs[idx[k]]  (or xs[t] = sum, "store").
//
// Work is organised as PHASES separated by workgroup barriers; in every phase each of the nw waves owns one
// (possibly empty) contiguous range of wave-steps (64 lanes, one slot per lane).  A step (descriptor layout:
// sched_format.h) lets groups of T = 2^lt lanes accumulate one target row each; on a flush step the groups are
// reduced and applied to their rows.  Long rows span several steps (flush on the last one).
//
// Triangular solves: per elimination level of <=16-row chunks of supernodes there are two phases,
//   A  rows of the level minus their couplings to earlier levels:  t_c = b_c - L[c, earlier] x   (subtract)
//   B  the in-chunk triangle, applied as a dense product with the INVERTED unit-lower diagonal block:
//      x_c = inv(L_cc) t_c  (store).  The inverse is part of the factor (Analysis::inv_off): a 15-deep
//      dependent chain inside one wave becomes <=16 independent 16-entry rows.  Inputs t and outputs x of
//      phase B live at different positions of the solve vector (Analysis::xloc), so the phase is race-free.
// Steps are numbered WAVE-MAJOR: all steps of wave 0 (phase after phase), then wave 1, ...  A wave therefore
// walks ONE linear stream per solve, its loads run a fixed number of steps ahead regardless of phase
// boundaries, and a phase boundary is nothing but a barrier count in the descriptor of the next step.
//
// Logical slot = physical slot = step*64 + lane; QP b's double of a slot sits at slot*BT + b inside a tile.
struct Schedule {
  int n_phases = 0, nw = 0, bt = 1;
  bool barriers = true;          // false: the check-SpMV schedule (independent rows, no barriers at all)
  std::vector<uint32_t> phase;  // diagnostics: per phase, stride 4*nw+1: kind (0 = A, 1 = B) then per wave (begin, end, 0, 0)
  std::vector<int> level_first_phase;   // first phase of every level (+ end)
  std::vector<uint32_t> step;   // per step: descriptor (sched_format.h)
  std::vector<uint32_t> step_ob;  // per step: first entry of outA on a flush
  std::vector<uint32_t> outA;   // target rows of flush steps (64/T each), kNoRow = none
  std::vector<uint32_t> idx;    // per slot: gather index into the solve vector
  std::vector<uint32_t> idxw;   // DEVICE index words, one per slot: low 16 bits = gather index, high 16
                                // bits = target row of the lane's group on flush steps (0xFFFF = none)
  std::vector<uint64_t> idxw64; // wide form (vectors of 65 535 entries and more; replaces idxw): low 32 bits =
                                // gather index, high 32 bits = target row (0xFFFFFFFF = none)
  std::vector<int32_t> src;     // per slot: canonical value index, MI_SRC_ZERO = structural zero, MI_SRC_ONE = 1.0
  uint32_t n_slots = 0;         // = n_steps * 64
  uint32_t n_steps = 0;
  std::vector<uint32_t> wave_range;      // 2 per wave: [begin, end) of its stream
  std::vector<uint32_t> lvl_pos;         // (n_levels+1) x nw: stream position of wave w at the start of level L
  std::vector<uint32_t> tail_bar;        // per wave: barriers still owed after its last step (all waves pass n_phases)
  // dataflow form (Analysis::df): no barriers; a subtracting flush reads the old value at `row` and writes the result at
  // `row + shadow`, every entry of the solve vector a step gathers is either written before the sweep starts or written
  // exactly ONCE during the sweep (the gatherer waits until it no longer holds the "not yet" pattern); padding slots
  // gather entry `pad` (always 0.0)
  bool dataflow = false;
  uint32_t shadow = 0, pad = 0;
  uint32_t phys_steps() const { return n_steps; }
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
  std::vector<uint32_t> tri4;         // 4/triple, resolved for the device (one 16-byte load instead of a chain of three):
                                      // offset of (I,K), offset of (J,K), first column of K, (h_I << 16) | (w_K << 8) | h_J
  std::vector<uint32_t> utask4;       // DEVICE form, 4/task (one 16-byte load, no second look-up): offset of the target block, tri_begin,
                                      // (h << 27) | (w << 22) | rank-1 triples, general triples
  bool overflow = false;              // a count that does not fit its field of the device form
  std::vector<uint32_t> ubig;         // 3 per level: first SMALL update task, first NARROW diagonal task, first NARROW solve task (<= 4 columns);
                                      // update tasks: first update task of the level's SMALL tasks (<= 16 triples, target <= 8 columns; they come last): [u_begin, ubig) one task per wave with its sources split over the lane groups, [ubig, u_end) one
                                      // task per 16-lane group (factor_kernel at one QP per workgroup)
  std::vector<uint32_t> dtask4;       // DEVICE form, 4/task: the blk tuple of the diagonal block
  std::vector<uint32_t> ttask4;       // DEVICE form, 4/task: offset of the block, first column of its chunk, (h << 8) | w, offset of the diagonal block
  std::vector<uint32_t> dtask;        // diagonal block ids
  std::vector<uint32_t> ttask;        // 2/task : block id, diagonal block id
  std::vector<uint32_t> asm_dst;      // per natural KKT entry: position in block storage
  std::vector<uint32_t> asm_src;      // per natural KKT entry: (kind << 29) | index
  std::vector<int32_t> lpos;          // canonical factor entry (L, then the inverted diagonal blocks) -> position in block
                                      // storage; inv(L_JJ)[i,k] (i > k) sits in the unused upper triangle of B(J,J), at (k,i)
  int n_levels = 0;
  size_t n_blocks() const { return blk.size() / 4; }
};
enum { ASM_P = 0, ASM_P_SIGMA = 1, ASM_SIGMA = 2, ASM_A = 3, ASM_NEG_RHOINV = 4 };

// Dense tail.  With a fill-reducing ordering of a KKT matrix whose graph is expander-like (random A) almost all
// of nnz(L) sits in one dense trailing triangle.  For the last k rows (k a multiple of 64, chosen by analyze())
// the factorisation then stops at the Schur complement S (k x k, dense, symmetric) and the solve applies
// M = S^-1 as ONE symmetric product between the forward and the backward sweep:
//     x_tail = M t_tail        instead of     L22 D2 L22' x_tail = t_tail   (two dependent triangular sweeps)
// M is streamed once per solve (k^2/2 values, every value used for row i AND for row j) where L22 was streamed
// twice (k^2 values), and the ~k/16 dependent levels of the trailing triangle become a handful of phases.
//
// Layout of the product: M is cut into 64 x 64 blocks (I >= J).  A wave owns a block for 64 steps; at step q lane l
// holds M[I0 + (l + q) % 64, J0 + l] ("circulant" order): the column partial sums stay in the lane, the row
// partial sums and the t values of the rows travel one lane per step (DPP wave rotate), so a block costs no LDS
// traffic until its two 64-vectors of partial sums are added to y.  Diagonal blocks use steps 1..32 only (each
// unordered pair once; the diagonal itself is a separate vector).  Blocks of one phase touch disjoint parts of
// the two accumulation vectors y_r / y_c (block (I, J): rows of I in y_r, rows of J in y_c, or swapped), phases
// are separated by workgroup barriers, x_tail = y_r + y_c.
struct DenseTail {
  int s = 0, k = 0, nb = 0;          // first permuted row, rows (multiple of 64; 0 = no dense tail), 64-row panels
  int nw = 0, n_phases = 0;
  uint32_t n_steps = 0;              // wave-steps of the value stream (64 slots each), wave-major
  std::vector<uint32_t> task;        // 4/task: I0, J0 (tail-local), flags (bit 0 diagonal block, bit 1 swap y_r / y_c) | barriers before << 8, steps
  std::vector<uint32_t> wave_task;   // nw + 1: task range of every wave
  std::vector<uint32_t> wave_step;   // nw + 1: stream range of every wave
  std::vector<uint32_t> tail_bar;    // nw: barriers owed after the last task (every wave passes n_phases)
  std::vector<int32_t> src;          // per slot: j * k + i (tail-local, i > j: column-major position in the dense k x k array) or MI_SRC_ZERO
  std::vector<uint32_t> sblk;        // k x k column-major, lower triangle incl. diagonal: block-storage position of S[i, j] (host replay)
  // ---- tables of tail_kernel (assembly of S on the matrix cores + blocked sweep inversion)
  // The k x k Schur complement lives as 16 x 16 TILES (I >= J, tile index I (I + 1) / 2 + J, 256 doubles each) in the register
  // order of v_mfma_f64_16x16x4_f64 accumulators: element (row, col) of a tile at ((row % 4) * 16 + col) * 4 + row / 4.
  // Assembly: S(I, J) = KKT block - sum over the columns c before the tail with entries in both tile rows of
  // L[I, c] d_c L[J, c]'.  The entries of L in tail rows / pre-tail columns ("compact entries", lt_*) are staged in LDS once
  // per QP; a QUAD is one MFMA step = 4 source columns: lane l (tile row / column l % 16, source l / 16) gathers compact
  // entry asm_q[quad][l] & 0xFFFF (operand A, tile row I) and >> 16 (operand B, tile row J); index n_lt = the zero slot.
  int n_lt = 0, n_ltcol = 0;
  std::vector<uint32_t> lt_pos;      // [n_lt]: block-storage position of the compact entry
  std::vector<uint32_t> ltcol_col;   // [n_ltcol]: permuted column of the compact source column (its D)
  std::vector<uint32_t> tile_tab;    // 4 / tile, in processing order (wave-major): (I << 16) | J, block-storage offset of the KKT block, first quad, end quad
  std::vector<uint32_t> wave_tiles;  // nw + 1: range of tile_tab entries of every wave
  std::vector<uint32_t> asm_q;       // [quads][64]
  std::vector<uint16_t> asm_qcol;    // [quads][4]: compact source column of the quad's 4 sources
  std::vector<uint64_t> asm_q64;     // DEVICE form of the two, [quads][64]: low word = byte offset of operand A's compact entry in LDS;
                                     // high word = byte offset of operand B's entry | (8 * compact source column) << 17
  std::vector<int32_t> src_tile;     // per stream slot: tile-order offset of its element (src translated) or MI_SRC_ZERO
  std::vector<uint32_t> diag_tile;   // [k]: tile-order offset of S[i, i]
  std::vector<uint32_t> task_step;   // per task (in stream order = the order of `task`): its first step in the value stream
  size_t asm_lds_bytes() const { return ((size_t)n_lt + 1 + (size_t)n_ltcol) * sizeof(double); }
};
// tile-order offset of element (i, j), i >= j (tail-local), of the lower-triangular tile array
inline uint32_t dt_tile_offset(int i, int j) {
  const int I = i / 16, J = j / 16, r = i % 16, c = j % 16;
  return (uint32_t)((I * (I + 1) / 2 + J) * 256 + ((r % 4) * 16 + c) * 4 + r / 4);
}
enum { DT_DIAG = 1, DT_SWAP = 2 };

struct Analysis {
  int n = 0, m = 0, N = 0;
  bool wide = false;               // n + m or 2n + m >= 65 535: 32-bit gather / row indices in the step streams (Schedule::idxw64)
  // triu(P) and A patterns (CSC, 32-bit on our side)
  std::vector<int> Pp, Pi, Psrc;   // Psrc: index into the caller's P value array
  std::vector<int> Ap, Ai;
  // natural upper-triangular KKT [[P+sigma I, A'],[A, -1/rho]]
  std::vector<int> Kp, Ki, PtoK, AtoK, rhotoK, sigmaOnlyK;
  std::vector<char> PisDiag;
  // fill-reducing permutation (perm[new] = old)
  std::vector<int> perm, pinv;
  int ordering = 0;                // 0 minimum degree, 1 nested dissection (picked by analyze())
  int relaxed_zeros = 0;           // > 0: relaxed supernodes with up to that many explicit zeros each (picked by analyze())
  // permuted KKT, lower triangle by columns, and map natural entry -> position
  std::vector<int> Klp, Kli, KtoKl;
  // symbolic factor: strictly-lower L by columns (sorted rows) + row view
  std::vector<int> Lp, Li, Rp, Rj, Rpos, etree;
  std::vector<int> sn_start;       // supernode boundaries
  std::vector<int> chunk_start;    // <=16-column chunks (phase-B blocks)
  std::vector<int> chunk_lev;      // forward level of every chunk
  // inverted diagonal blocks: chunk c with r >= 2 columns owns r(r-1)/2 canonical values after the nnz(L)
  // entries of L: entry (i,k), i > k (local), at nnzL() + inv_off[c] + k*(2r-k-1)/2 + (i-k-1); inv_off = -1 for r = 1
  std::vector<int> inv_off;
  int n_inv = 0;
  // position in the solve vector of the FORWARD result / BACKWARD input of permuted row j: j itself for
  // one-row chunks, N + (running index) for the rows of multi-row chunks (Next = vector length)
  std::vector<int> xloc;
  int Next = 0;
  // Dataflow form of the triangular solves (tri_waves > 0 of analyze(): ONE QP shared by that many waves, solve
  // vector in global memory).  The waves do not meet at barriers between the levels: a value that a sweep produces is
  // written once, to an entry that holds the "not yet" bit pattern until then, and its consumers poll for it.  For that
  // the in-place results get a second home: entry e + Next ("shadow") for a row whose sweep step subtracts in place.
  //   forward : A rows read the right-hand side at i and write t_i at i + Next; B rows write at xloc[i]
  //   backward: A rows read xloc[k] and write at xloc[k] + Next; B rows write at k
  // rflag[e]: bit 0 = row e has a forward A step, bit 1 = a backward A step, bit 2 = row of a multi-row chunk.
  // Entry 2 * Next is the padding entry (0.0); the vector has xs_total entries.
  bool df = false;
  int tri_waves = 0;
  int xs_total = 0;
  std::vector<uint8_t> rflag;
  // where the forward / backward sweep leaves the final value of row e
  int df_floc(int e) const { return (rflag[e] & 4) ? xloc[e] : ((rflag[e] & 1) ? e + Next : e); }
  int df_bloc(int e) const { return (rflag[e] & 4) ? e : ((rflag[e] & 2) ? e + Next : e); }
  Schedule fwd, bwd, chk;
  BlockFactor bf;
  DenseTail dt;                    // dt.k == 0: none
  std::vector<int32_t> fwd_srcblk, bwd_srcblk;   // fwd/bwd slot -> block-storage position (-1 = zero)
  int nnzL() const { return Lp.empty() ? 0 : Lp.back(); }
  int nnzLx() const { return nnzL() + n_inv; }     // length of the canonical factor array QPNumeric::Lx
  int inv_index(int c, int i, int k) const {       // canonical index of inv(L_cc)[i,k], i > k local
    const int r = chunk_start[c + 1] - chunk_start[c];
    return nnzL() + inv_off[c] + k * (2 * r - k - 1) / 2 + (i - k - 1);
  }
  int nnzK() const { return Kp.empty() ? 0 : Kp.back(); }
};

// E1 (pattern part), E4, E5-symbolic and the schedules.  Returns 0 or an error
// code of include/mi_osqp.h.
// `nwaves` = waves per workgroup the device kernels will run with, `bt` = QPs per
// tile (the step streams are laid out per wave; bt only sizes the physical layout).
// `max_extra_rows` = how many rows may get a second position in the solve vector (Analysis::xloc): the caller's
// LDS capacity; negative = no limit of the caller's.  With 16-bit index words (Analysis::wide == false) the count
// is also limited by the index range.
// `dense_tail_max` = largest dense tail (rows) the caller can serve, 0 = never use one.
// `tri_waves` > 0 = the dataflow form of the triangular solves (Analysis::df: ONE QP, solve vector in global memory, no
// dense tail): the forward / backward step streams and the check schedule are laid out for that many waves (of any number
// of workgroups; a kernel with fewer waves walks the barrier-free check streams one after the other).  0 = the barrier form
// (one workgroup of nwaves waves per tile).
int analyze(int64_t n, int64_t m, const int64_t *Pp, const int64_t *Pi, const int64_t *Ap,
            const int64_t *Ai, Analysis &an, int nwaves = 8, int bt = 1, int max_extra_rows = -1,
            int dense_tail_max = 512, int tri_waves = 0, int n_tiles = 1);
// `n_tiles` = tiles of the batch the analysis is for (they share the device's memory bandwidth: the decision for relaxed
// supernodes weighs fewer phases against a longer factor stream).
// physical position (in doubles, inside one tile) of QP b's value of a logical slot
size_t phys_index(const Schedule &s, uint32_t slot, int b);

// Per-QP numeric state kept on the host (needed for rescaling and refactors).
struct QPNumeric {
  std::vector<double> Pv, Av, q, l, u;            // scaled problem data
  std::vector<double> D, Dinv, E, Einv;           // Ruiz scaling
  double c = 1.0, cinv = 1.0;
  double rho = 0.1;
  std::vector<double> rho_vec, rho_inv;
  std::vector<int8_t> ctype;
  std::vector<double> Lx, Dl, Dlinv;              // factor: L in canonical CSC order, then the inverted diagonal blocks (Analysis::nnzLx)
  std::vector<double> Minv;                       // dense tail: S^-1, k x k column-major (both triangles)
};

void load_qp(const Analysis &an, const Settings &st, const double *Pval, const double *q,
             const double *Aval, const double *l, const double *u, QPNumeric &qp);
void scale_qp(const Analysis &an, const Settings &st, QPNumeric &qp);       // E2
void scale_like(const Analysis &an, const QPNumeric &rep, QPNumeric &qp);   // same unscaled P, A, q as rep: reuse its scaling
void unscale_qp(const Analysis &an, QPNumeric &qp);
void set_rho_vec(const Analysis &an, const Settings &st, QPNumeric &qp);    // E3
// returns 1 if any constraint type changed (bounds update path of E13)
int refresh_rho_types(const Analysis &an, QPNumeric &qp);
void apply_rho(const Analysis &an, QPNumeric &qp, double rho_new);
// E5 numeric: left-looking LDL' on the permuted KKT. 0 ok, MI_OSQP_ERR_NONCONVEX else.
int factor_qp(const Analysis &an, const Settings &st, QPNumeric &qp, std::vector<double> &work);
// reference solve with the canonical factor (natural order in/out)
void direct_kkt_solve(const Analysis &an, const QPNumeric &qp, const double *rhs, double *sol);
// x_tail = M t_tail exactly as the device evaluates it (task order, fma order); false: two tasks of a phase collide
bool replay_dense_tail(const DenseTail &dt, const double *M, const double *mdiag, double *xt);
// sequential interpreter of the device schedules (tests only; see mi_osqp.h)
bool replay_kkt_solve(const Analysis &an, const QPNumeric &qp, const double *rhs, double *sol);   // false: the streams are not race- / deadlock-free
// host interpreter of the device block factorisation (tests only): fills Lx/Dlinv
// of `out` from qp's scaled data and rho vector exactly as factor_kernel does
int replay_block_factor(const Analysis &an, const Settings &st, const QPNumeric &qp, QPNumeric &out);
// combined value array the check-SpMV schedule indexes: [P triu | A]
void replay_spmv(const Analysis &an, const QPNumeric &qp, const double *x, const double *y,
                 double *Px, double *Aty, double *Ax);

}  // namespace miosqp
