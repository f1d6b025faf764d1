// gdyn_types.h -- structures shared by the host side (gdyn_capi.hip) and the kernels
// (gdyn_kernels.hip).  Layout in HBM (DESIGN.md "Data layout"):
//   * beads of one replica live in SLOT order: slots are re-assigned by a counting sort
//     over the neighbour-search cells at every list build, so that spatial neighbours are
//     memory neighbours (coalesced loads, L1/L2-local gathers).  orig[] / slot_of[] map
//     between slot order and the caller's bead (chain) order.
//   * every per-slot array is replica-major with stride Np (N padded to the block size):
//     element (replica r, slot s) is at r*Np + s.
//   * pair lists and bond adjacency are stored in 16-byte chunks, wave-interleaved: chunk c of slot g is
//     uint4 #((g/64)*NC + c)*64 + g%64 -- one coalesced 1 KiB read per wave per chunk.
//   * meta[g] = bond degree | point-source mask << 8 | pair-list length << 16.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef GD_BLOCK
#define GD_BLOCK 512
#endif
#define GD_MAX_BOND_TYPES 16   // distinct bond parameter sets (the reference models use <= 4 + the loop / glue slots); the
                               // table lives in LDS next to the tile: every 16 bytes saved there are one more tile entry
#define GD_MAX_POINT_SOURCES 4
#define GD_ADJ_SHIFT 26                    // bond adjacency entry = partner | type << 26 (| GD_ADJ_LOCAL)
#define GD_ADJ_MASK ((1u << GD_ADJ_SHIFT) - 1u)
#define GD_ADJ_LOCAL 0x80000000u           // tiled path: partner field is an index into the block's LDS tile
#define GD_CHAIN_LOCAL 0x40000000          // same for chain (bending) partners
#define GD_TILE_RANGES 9                   // (dz,dy) rows of the 27-cell neighbourhood
#define GD_XCDS 8
#define GD_REC_NOBEAD 0xffffffffu          // tiled per-thread record: rec_mo.y of a thread without a bead
// tiled per-thread record (rec_mo): x = bond degree | point-source mask << 8 | block-local slot << 12 | near entries / 4 << 21 (11 bits),
// y = bead id (26 bits, like a bond partner) | far chunks << 26 (6 bits).  The classes are capped where the fields end.  A bead whose far
// class does not fit (the spline-refined start of the pipeline: some beads have 4 000 neighbours inside the default list radius) is
// flagged (GD_FLAG_OVERFLOW bit 1) and the host builds single-class lists -- everything near -- until the dense state has passed.
#define GD_TILED_MAX_NEAR 8184u            // near entries (1 023 chunks of 8; 2 046 fours in the record's 11 bits)
#define GD_TILED_MAX_FAR 504u              // far entries (63 chunks)
#define GD_TILED_MAX_W (GD_TILED_MAX_NEAR + GD_TILED_MAX_FAR)
#define GD_REC_ID_MASK 0x03ffffffu         // bead id field of rec_mo.y; all ones = no bead
#define GD_DMAX_STRIDE 32u                  // words between the replicas' displacement maxima: one 128-byte line each
#define GD_REPAIR_GRID 1024u               // blocks of the repair launch behind every k_fill (one k_step wave each; they leave at once while the
                                           // queue is empty); a build that queues more is flagged, and the host launches one block per wave
#define GD_UNROLL 8u                       // pair-list batch: lists are padded to a multiple of this

enum { GD_MODE_STEP = 0, GD_MODE_FORCE = 1, GD_MODE_ENERGY = 2 };

// flags[r*GD_NFLAGS + k]
enum { GD_FLAG_VIOLATION = 0, GD_FLAG_OVERFLOW = 1, GD_FLAG_MAXDISP2 = 2, GD_FLAG_NEED_W = 3, GD_FLAG_TILE_OVERFLOW = 4,
       GD_FLAG_NEED_TILE = 5, GD_FLAG_TAINT = 6, GD_FLAG_NCELL = 7, GD_NFLAGS = 8 };
// GD_FLAG_OVERFLOW bits: 1 a row is too narrow for its list, 2 a list class beyond its field of the tiled record, 4 the row pool is full
// GD_FLAG_NCELL: cells of the replica's grid at the last build (sizes the scan's launch at the next one)
// GD_FLAG_TAINT: set by a list build that starts after an overflow was flagged in the same chunk -- the steps since ran on
// incomplete lists, the positions are no basis for sizing anything: such a build reports no needs (the chunk is rolled back)

// LDS tile of one block (tiled path): the block's 256 slots plus the slots of every cell adjacent
// to its cells, as 9 contiguous slot ranges (one per (dz,dy) row offset of the cell grid).
struct TileDesc {   // all fields 32-bit: the kernels read it through a block-uniform pointer with scalar (s_load) loads;
                    // 16-bit fields would be fetched with per-lane vector loads and a full vmcnt wait each
    unsigned start[GD_TILE_RANGES];        // first slot of the range
    unsigned len[GD_TILE_RANGES];          // slots in the range
    unsigned base[GD_TILE_RANGES];         // LDS index of the range's first slot
    unsigned total;                        // beads staged
    unsigned nranges;                      // merged ranges in use (0: the tile did not fit)
    unsigned own_base;                     // LDS index of the block's first own slot (blk * GD_BLOCK): k_step reads its beads from the tile
    unsigned pad_[2];
    // per (dz,dy) row offset k: first slot of cell c0+off_k-1 and its LDS index (0xffffffff: no such row)
    unsigned kstart[GD_TILE_RANGES];
    unsigned kbase[GD_TILE_RANGES];
};
#define GD_TD_TOTAL 27
#define GD_TD_NRANGES 28
#define GD_TD_OWN 29

struct DevCtx {                 // per replica, fp64 (a few scalars; kept exact)
    long long step;
    int pending;                // 1: the callback of step `step+1` has not been applied yet
    int pad;
    double time, bead_scale, bond_scale;
    double semi[3];
    double react[3];            // axial_reaction of the last force evaluation
};

struct CtxF {                   // float copy of DevCtx + block-uniform wall constants; one per block in LDS
    long long step;
    float bead_scale, bond_scale, semi[3];
    float inv_semi[3], inv_semi2[3];
    float inv_bond_scale2;
    float near2;                // a bead can touch the wall only if C + 1 >= near2 (see the wall section); <= 0: always
    float w_inv_sa2, w_inv_sb2, w_ca, w_cb;     // soft wall (half diameters, bead scale): 1/s^2, 6 eps_a / sa^2, 24 eps_b / sb^2
    float p_inv_sa2, p_inv_sb2, p_cut;          // pair potential at the current bead scale: 1/sigma^2 of both cores, cutoff
    float sg_uniform;                           // sqrt(2 mu kT dt) for the uniform mobility (< 0: per-bead mobilities)
    // the wall's level-set value C = x^2/a^2 + y^2/b^2 + z^2/c^2 - 1 without its cancellation (beads in reach of the wall have
    // |C| < 0.05: in fp32 the difference of two numbers near 1 loses three digits): numerator x^2 b^2c^2 + y^2 a^2c^2 + z^2 a^2b^2 -
    // a^2b^2c^2 in fp64 from the fp64 semiaxes, times the fp32 reciprocal of a^2b^2c^2
    float w_inv_q3;                             // 1 / (a^2 b^2 c^2)
    double w_q[4];                              // b^2c^2, a^2c^2, a^2b^2, a^2b^2c^2
};

struct GridP {                  // per replica cell grid of the last list build
    float org[3];
    float inv[3];               // 1 / cell size
    int nc[3];
    int ncell;
};

struct PairP {
    float eps_a, sigma_a, eps_b, sigma_b;
    int p_a, q_a, p_b, q_b;
    int mix, scaled, enabled;
    float cutoff;               // max(sigma with eps != 0), unscaled
};

struct WallP {
    float eps_a, sigma_a, eps_b, sigma_b;
    int p_a, q_a, p_b, q_b;
    float wall_a, wall_b;
    int scaled, enabled;
    int fast2383;               // the wall's soft cores are <2,3> + <8,3>: branch-free specialisation
    float packing_spring;
    double spring[3], mobility;
    // inner spherical wall (gd_set_inner_sphere_wall)
    int inner_enabled;
    float in_radius, in_eps_a, in_sigma_a, in_eps_b, in_sigma_b, in_wall_a, in_wall_b, in_spring;
    int in_p_a, in_q_a, in_p_b, in_q_b;
};

struct BondType {               // 32 bytes: a 16-byte and an 8-byte LDS read per bond (kind / pq only on the rare soft-core path)
    float ka, kb, la, lb;
    int flags;                  // mix | scaled << 1 | minimg << 2 | term << 8
    float xmin;                 // lower clamp of the elongation r - l: 0 for the semispring, -inf otherwise
                                // (harmonic = spring with l = 0, so one branch-free form covers the three)
    int kind;
    int pq;                     // p | q << 8 (GD_POT_SOFTCORE)
};

struct PointSrc {
    int kind;
    float k, b, p[3];
};

struct ScaleP {
    int enabled;
    int from_host;              // 1: every replica is at the same step, so the scales the pending callback sets were computed by the host
                                //    (two fp64 exponentials less on the critical path of every block's first wave)
    double bead_init, bead_tau, bond_init, bond_tau;
    double bead_next, bond_next;       // bead_scale / bond_scale of the step the pending callback moves to (from_host)
};

struct StepParams {
    // sizes
    unsigned N, Np, R, nblk;            // nblk = blocks per replica
    size_t stride;                      // R*Np, ELL column stride
    int periodic;
    float box[3], inv_box[3];
    // per-slot state
    const float4 *pos_in;
    float4 *pos_out;
    const float4 *xb;                   // positions at list build
    const unsigned *orig;
    const float2 *ab;
    const float *mob;
    const float4 *bendE;                // (e_last, e_mid, e_first, -) bending energies of the 3 triplets a bead is in
    // lists
    const unsigned *nbr;
    const unsigned short *nbr16;        // tiled path: tile-local indices
    const TileDesc *tiles;              // [R][nblk]
    const uint2 *wtab;                  // tiled lists: [R * Np / 64] per k_step WAVE, x = first KiB of the wave's rows in nbr16 (a KiB = one
                                        // 16-byte chunk of 64 lanes), y = chunks per lane of those rows (ragged rows: every wave has its own width)
    const unsigned *meta;               // bdeg | psmask << 8 | list length << 16
    unsigned W, WB;                     // generic lists: row width (entries); bond adjacency width (entries, multiple of 4)
    float mob_uniform;                  // >= 0: every bead has this mobility (mob[] is not read)
    int tiled;                          // 1: LDS-tiled path
    int pk;                             // 1: softcore<2,3> + softcore<8,3> specialisation
    int packed_ab;                      // 1: pos.w holds (a,b) as two fp16 (exactly representable)
    unsigned cpb;                       // blocks per replica per XCD (XCD-aware block mapping)
    unsigned tile_cap;                  // beads of LDS per block
    unsigned tile_lo, tile_hi;          // tile_hi != 0: this launch runs only the blocks whose tile holds tile_lo < beads <= tile_hi (the
                                        // step is then two launches: the tiles that fit the three-block LDS class, and the few larger ones)
    // tiled path: per-THREAD records written by the build (threads of a block are ordered by list length so that the
    // lanes of a wave run the same number of list batches): build position + block-local slot, meta + bead id
    const float4 *rec_x0;
    const uint2 *rec_mo;
    int has_softcore_bonds;             // some bond set is a soft core (the glue pairs of the 1 kb model): rare path
    int bonds_premixed;                 // no bond record asks for AB mixing (the host resolved it per bond): K = ka, l = la
    int bonds_all_scaled;               // every bond record scales with bond_scale (the genome models): no per-bond select
    const unsigned *badj;               // chunked like the pair lists, 4 entries per chunk
    const int4 *chain;                  // slots of (i-2, i-1, i+1, i+2) or -1
    // context
    const DevCtx *ctx_in;
    DevCtx *ctx_out;
    const float4 *react_in;             // [R][nblk] wall-reaction partials of the previous step (read by the callback)
    float4 *react_out;                  // [R][nblk] this step's partials (double-buffered with the context)
    unsigned *flags;
    // model
    PairP pair;
    WallP wall;
    ScaleP scaling;
    const BondType *btab;
    int nbt;
    PointSrc ps[GD_MAX_POINT_SOURCES];
    int nps;
    int has_bend, has_bonds;
    // run
    double dt_d;
    float dt, kT;
    unsigned long long seed;
    const unsigned long long *seeds;    // [R] per-replica seeds (gd_run_desc.replica_seeds) or NULL
    int noise_mode, run_flags;
    const float *host_noise;            // (R, N, 3) normals of this step
    float rv;                           // list radius (violation check)
    float rn;                           // near-class radius of the tiled list in use
    unsigned *dmax;                     // [R] largest squared displacement since the build, float bits (monotone between builds)
    int record_disp;
    // compensated position update (small-dt / T = 0 runs: mu F dt below the ulp of an fp32 coordinate): the true position of a bead is
    // pos + lo with lo the fp32 residual the rounded sums left behind, [R][N] by BEAD index (it does not take part in the cell sort)
    float4 *lo;
    int comp;                           // 1: x += e as a two-sum over (pos, lo)
    // force / energy modes
    unsigned term_mask;
    float4 *fout;                       // [R][N] by bead index
    double *epart;                      // [R][nblk]
};

struct BuildParams {
    unsigned N, Np, R, nblk;
    size_t stride;
    int periodic;
    float box[3], inv_box[3];
    float rv;
    float rn;                           // near-class radius of tiled lists (cutoff < rn <= rv)
    unsigned *dmax;                     // [R] largest squared displacement since the build (float bits)
    unsigned ncell_cap;
    unsigned scan_segments;             // blocks per replica of k_scan (the last one walks whatever is left)
    int kx;                             // open boxes: cells are 1/kx of the list radius wide in x (a bead's row window is 2 kx + 1 cells =
                                        // (2 + 1/kx) radii instead of 3); periodic grids: 1
    const float4 *pos_in;               // current order
    float4 *pos_out;                    // new (sorted) order
    float4 *xb;
    const unsigned *orig_in;
    unsigned *orig_out;
    unsigned *slot_of;                  // [R][N]
    unsigned *rank;                     // [R*Np] position of the bead inside its cell as the atomics of k_bin handed it out (arrival order)
    unsigned *members;                  // [R*Np] bead ids by (cell, arrival rank) (k_members): k_scatter ranks a bead by its ID inside its cell,
                                        // so the slot order -- and with it every fp32 summation order downstream -- is a function of the
                                        // positions alone, not of the order in which the atomics of k_bin happened to arrive
    unsigned *cell_cnt, *cell_start;    // [R][ncell_cap+1]; the counters are zero between builds (k_fill clears the cells a build used)
    float *bbox;                        // [R][nblk][6] per-block bounding-box partials (k_bbox: builds without a bounding box from the build before)
    // Open boxes: the bounding box of the positions a build sorted (k_scatter: one partial per wave, reduced by the extra blocks of
    // k_tiles) is the box the NEXT build lays its grid on -- beads move less than the skin in between and cell_coords clamps to the
    // grid, so any box gives correct lists, a stale one slightly fuller boundary cells; the first build of a handle, every build after
    // gd_set_positions or a rolled-back chunk, and builds of generic lists (no k_tiles) run k_bbox + k_gridp instead.
    const float *bbox_cur;              // [R][6] lo[3], hi[3] (warm builds read it)
    float *bbox_next;                   // [R][6] written by k_tiles
    float *bbox_w;                      // [R][nblk * GD_BLOCK / 64][6] per-wave partials of k_scatter
    int warm;                           // 1: k_bin lays the grid itself (from bbox_cur, or from the periodic box); 0: k_bbox + k_gridp ran first
    GridP *grid;
    // static per-bead (bead order)
    const float2 *ab_o;
    const float *mob_o;
    const float4 *bendE_o;
    const unsigned char *psmask_o;
    const unsigned *badj_o;             // [WB][N]
    const unsigned char *bdeg_o;
    const int4 *chain_o;
    unsigned WB;
    // slot-order outputs
    float2 *ab;
    float *mob;
    float4 *bendE;
    unsigned *badj;
    int4 *chain;
    unsigned *nbr, *meta;
    int has_bend, mob_is_uniform;
    unsigned short *nbr16;
    TileDesc *tiles;
    unsigned W;
    int tiled, packed_ab;
    int w_valid;                        // pos_in.w already holds the packed (a,b) (written by an earlier build, kept by every step)
    unsigned cpb, tile_cap;
    unsigned *flags;
    unsigned long long *lcount;         // [2 R] directed list entries per replica, then (tiled lists) their near entries in fours
    float4 *rec_x0; uint2 *rec_mo;      // per-thread records of the tiled path (see StepParams)
    unsigned char *len_prev;            // [R*N] by bead id: list batches at the previous build (the balancing sort key)
    // Ragged rows of the tiled lists.  The rows of one k_step wave (64 threads, ordered by list length: near-uniform lists) are as
    // wide as the wave's longest list needs; a build takes them from one pool (nbr16) with a bump cursor, one atomic per block.
    // The width has to be known before the first entry is written: it is PREDICTED from what each bead needed at the build before
    // (need_prev, + an eighth and a chunk to spare per class; without history: p.W entries).  A list that outgrows its wave's rows
    // is repaired: rows keep counting past their width, so the exact need is known when the block ends -- its last wave queues the
    // k_step waves concerned, and the repair kernel behind k_fill (k_fill<..., REPAIR>: one wave per queue item) takes fresh rows of
    // that width from the pool and lists the 64 beads of the wave again.  No rollback; the queue is empty in almost every build.
    // Only a FULL POOL (or queue) is the host's business: GD_FLAG_OVERFLOW bit 4; the cursor keeps counting -- its final value is the need.
    uint2 *wtab;                        // [R * Np / 64] (first KiB, chunks per lane) per k_step wave
    unsigned short *need_prev;          // [R*N] by bead id: near chunks (10 bits) | far chunks << 10 (6 bits) the last build counted
    unsigned *pool;                     // [0] cursor (KiB taken so far; its final value is the pool's use, or the need when it was full; k_scan
                                        // starts it over), [1] the largest final value of the builds before this one, [2] k_step waves
                                        // repaired -- [1] and [2] since the host last cleared them (with the flags) --, [3] items in the
                                        // repair queue of this build (k_scan starts it over)
    uint2 *rqueue;                      // repair queue: x = (replica * nblk + block) << 3 | k_step wave of the block, y = chunks per lane needed
    unsigned rq_cap;                    // items the queue holds: every k_step wave of the handle
    unsigned rq_grid;                   // blocks of the repair launch (GD_REPAIR_GRID; rq_cap while a fast-changing state needs more)
    unsigned pool_cap;                  // KiB of the pool
    int predict;                        // 1: need_prev describes these beads (the build before ran at this radius and class mode)
    unsigned long long *dbg;            // section stamps of timing-only builds (the force-output buffer)
};

// launchers (gdyn_kernels.hip)
hipError_t gd_kernels_init_device(void);      // LDS opt-in of every kernel that needs it, on the current device (once per device: gd_create)
void gd_launch_step(const StepParams &p, int mode, hipStream_t st);
void gd_launch_finalize(const StepParams &p, int mode, hipStream_t st);     // k_ctx: 0 final callback, 1 fold reaction partials
void gd_launch_build(const BuildParams &p, hipStream_t st);
// Droplet attraction among a small set of target beads (gd_set_pair_softwell): all pairs, one thread per target.
struct SoftwellP {
    const float4 *pos_in;       // positions the forces are evaluated on (slot order)
    float4 *pos_out;            // mode 0: x_out += mu dt F  (the Euler-Maruyama update is linear in F)
    float4 *fout;               // mode 1: forces by bead index, added
    double *esum;               // mode 2: [R] energy, added
    const unsigned *slot_of, *targets;
    const float *mob_o;         // per-bead mobility, or NULL with mob_uniform
    float mob_uniform, dt, eps, inv_d2, rc2;
    float4 *lo;                 // mode 0 with comp: residuals of the compensated update, [R][N] by bead (k_step has normalised (x, lo) already)
    int comp;
    unsigned N, Np, R, M;
    int periodic;
    float box[3], inv_box[3];
};
void gd_launch_softwell(const SoftwellP &p, int mode, hipStream_t st);
// Unique pairs (bead ids i < j) of one replica closer than dcut, filtered from the RESIDENT Verlet list (k_pairs)
struct PairsP {
    const float4 *pos, *x0;             // current positions (slot order); build positions: rec_x0 (tiled, thread order) or xb (slot order)
    const uint2 *rec_mo; const unsigned *meta, *orig;
    const unsigned *nbr; const unsigned short *nbr16; const TileDesc *tiles; const uint2 *wtab;
    unsigned N, Np, nblk, r, W;
    int tiled, s16, periodic;
    float box[3], inv_box[3];
    float dcut2, lim2;                  // lim2: largest squared displacement since the build for which the list still holds every pair within dcut
    uint2 *out; unsigned long long cap; // replica r + y of a multi-replica launch writes at out + y * cap
    unsigned long long *count;          // per replica of the launch: [2y] pairs found, [2y + 1] != 0: some bead has moved too far (the caller rebuilds)
    unsigned nrep;                      // replicas r .. r + nrep - 1 in one launch (grid.y)
    const unsigned *dmax;               // [R] x GD_DMAX_STRIDE: k_step's running bound of the squared displacement since the build (float
                                        // bits), or NULL: a replica whose bound already exceeds lim2 is flagged by its first wave and skipped
};
void gd_launch_pairs(const PairsP &p, hipStream_t st);

// Time-integrated contact maps (gd_contacts_*; simulation_interphase/contact_map.cc:31-91): one open-addressing table per replica,
// one 64-bit word per slot = (i << jbits | j) << cbits | count (bead ids i < j < 2^jbits, cbits = 64 - 2 jbits >= 24 count bits;
// all ones = empty slot: no pair has i all ones), so that counting a pair that is already in the table is ONE random access: a
// load that finds the key and a 64-bit atomic add on the same word.
#define GD_CT_EMPTY 0xffffffffffffffffull
struct ContactTab {
    unsigned long long *words;          // [R][cap]
    unsigned *distinct;                 // [R] occupied slots
    unsigned long long cap;             // slots per replica, a power of two
    unsigned jbits;                     // bits of a bead id
};
// count the pairs of `pairs` (cap_pairs per replica, count[2y] valid) into the tables of all R replicas
void gd_launch_contacts_insert(const ContactTab &t, const uint2 *pairs, unsigned long long cap_pairs, const unsigned long long *count,
                               unsigned long long max_count, unsigned R, hipStream_t st);
void gd_launch_contacts_rehash(const ContactTab &from, const ContactTab &to, unsigned R, hipStream_t st);      // (to: cleared, distinct zeroed)
// occupied slots of replica r -> keys_out (i << jbits | j) / vals_out (any order), *n_out = their number
void gd_launch_contacts_compact(const ContactTab &t, unsigned r, unsigned long long *keys_out, unsigned *vals_out, unsigned *n_out, hipStream_t st);
// gdyn_sort.hip: ascending radix sort of (key, value) pairs (rocPRIM); tmp == nullptr: only *tmp_bytes is set
hipError_t gd_sort_contacts(void *tmp, size_t *tmp_bytes, const unsigned long long *kin, unsigned long long *kout, const unsigned *vin,
                            unsigned *vout, size_t n, unsigned key_bits, hipStream_t st);
void gd_launch_gather_xyz(const float4 *pos, const unsigned *slot_of, float *out, unsigned N, unsigned Np, unsigned R, int quantize,
                          hipStream_t st);
void gd_launch_gather_positions(const float4 *pos, const unsigned *slot_of, float4 *out, unsigned N, unsigned Np,
                                unsigned R, int quantize, hipStream_t st);
void gd_launch_identity(unsigned *orig, unsigned *slot_of, unsigned N, unsigned Np, unsigned R, hipStream_t st);
