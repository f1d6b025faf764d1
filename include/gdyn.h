/* gdyn.h -- C-ABI of libgdyn: the MI355X Brownian-dynamics stepper.
 *
 * This is the one boundary the build cuts through the reference
 * (snsinfu/2022a-genome-dynamics): the set of micromd (<md.hpp>) calls that the
 * reference's simulation drivers make.  Every entry point below cites the
 * reference call site (path:line relative to the reference root) it replaces.
 * Reference potentials are arbitrary C++ lambdas; they cannot cross a C-ABI, so
 * the closed set of parameterised families the drivers actually instantiate is
 * exposed instead.
 *
 * Conventions
 *   - plain pointers and sizes only; all host arrays are caller-owned.
 *   - positions/forces on the boundary are fp64 (md::point is 3 doubles,
 *     3-sim-1kb/src/simulation/buffer_traits.hpp:15-30), shape (R, N, 3),
 *     replica-major; the device computes in fp32.
 *   - every function returns a gd_status; gd_last_error() gives the message.
 *     No C++ exception crosses the boundary.
 *   - a handle is bound to one HIP device and one stream; it is not
 *     thread-safe; independent handles may live on different threads/devices.
 *   - "bead index" is the index in the caller's (chain) order, always.
 */
#ifndef GDYN_H
#define GDYN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gd_system gd_system;

typedef enum {
    GD_OK = 0,
    GD_EINVAL = 1,       /* bad argument (index out of range, NULL, unsupported power ...) */
    GD_ENODEVICE = 2,    /* no HIP device / HIP extension unusable */
    GD_EHIP = 3,         /* a HIP runtime call failed */
    GD_ENOMEM = 4,
    GD_ESTATE = 5,       /* call sequence error (e.g. topology change after first run) */
    GD_EUNSUPPORTED = 6  /* e.g. spacestep != 0 (adaptive step; micromd-internal rule unknown) */
} gd_status;

const char *gd_last_error(void);
/* "hip" for the product library, "oracle" for oracle/liboracle (tests only). */
const char *gd_backend_name(void);

/* ------------------------------------------------------------------ system */

enum { GD_BOX_OPEN = 0, GD_BOX_PERIODIC = 1 };

typedef struct {
    uint32_t n_beads;      /* N: md::system::add_particle() count
                              (5-sim-genome/src/simulation_interphase/simulation_driver_particles.cc:17-20) */
    uint32_t n_replicas;   /* R independent trajectories stepped in one launch (new: the
                              reference runs one process per seed, 5-sim-genome/scripts/run_simulation:8-25) */
    int32_t  device;       /* HIP device ordinal */
    int32_t  box_kind;     /* GD_BOX_OPEN | GD_BOX_PERIODIC (md::periodic_box,
                              3-sim-1kb/src/simulation/simulation.cpp:118-123) */
    double   box[3];       /* x/y/z period when periodic */
} gd_desc;

/* Version of this header's struct layouts (gd_desc, gd_context, gd_run_desc, gd_tuning, gd_timing ...); bumped whenever one
 * of them changes.  A caller compiled against another version would have the library read or write past its structs, so
 * handle creation carries the caller's version and fails with GD_EINVAL on a mismatch: gd_create() is a macro over
 * gd_create_abi().  Bindings that do not go through this header (ctypes) pass their own constant. */
#define GD_ABI_VERSION 5
int gd_abi_version(void);
int gd_create_abi(int abi_version, const gd_desc *desc, gd_system **out);
#define gd_create(desc, out) gd_create_abi(GD_ABI_VERSION, (desc), (out))
int gd_destroy(gd_system *sys);

/* md::system::view_positions() (simultion_driver_relaxation.cc:12-15). (R,N,3) fp64. */
int gd_set_positions(gd_system *sys, const double *xyz);
int gd_get_positions(gd_system *sys, double *xyz);
/* Snapshot path: (R,N,3) fp32; quantize != 0 rounds to multiples of 2^-16 on the
 * device exactly as simulation_common/simulation_store.cc:257-268,403-407 does on the host. */
int gd_get_positions_f32(gd_system *sys, float *xyz, int quantize);

/* Per-bead attributes, shared by all replicas; any pointer may be NULL (keeps default).
 *   a_factor,b_factor: particle_data (simulation_common/particle_data.hpp:6-13), default 0,0
 *   mobility:          view_mobilities() (simulation_driver_particles.cc:25-35), default 1
 *   bending_energy:    monomer_data (3-sim-1kb/src/simulation/simulation.cpp:21-26,66-79), default 0 */
int gd_set_bead_params(gd_system *sys, const double *a_factor, const double *b_factor,
                       const double *mobility, const double *bending_energy);

/* ------------------------------------------------------- non-bonded (a3,a5) */

/* a*softcore<PA,QA>{eps_a,sigma_a} + b*softcore<PB,QB>{eps_b,sigma_b} over a Verlet
 * neighbour list; neighbour distance = max(sigma) (x bead_scale when scaled).
 *   mix=1: a=(a_i+a_j)/2, b=(b_i+b_j)/2  (simulation_driver_forcefield.cc:30-45,
 *          4-sim-ab/box/src/simulation/simulation_driver.cc:107-113)
 *   mix=0: a=b=1                          (3-sim-1kb/src/simulation/simulation.cpp:102-124,
 *          simulation_spindle/simulation_driver.cc:94-102 with eps_b=0)
 * softcore<P,Q>: U = eps*(1-(r/sigma)^P)^Q for r<sigma, P even in {2,4,6,8,12}, Q in {1,2,3,4}.
 * eps may be negative (attraction). Periodic systems use the minimum image. */
typedef struct {
    double  eps_a, sigma_a;
    double  eps_b, sigma_b;
    int32_t p_a, q_a, p_b, q_b;
    int32_t mix;
    int32_t scale_by_bead_scale;   /* sigma *= bead_scale(t), simulation_driver_forcefield.cc:37,41,48 */
} gd_pair_softcore;

int gd_set_pair_softcore(gd_system *sys, const gd_pair_softcore *p);

/* Nucleolar droplet attraction (simulation_driver_forcefield.cc:153-178, only added when
 * nucleolus_droplet_energy != 0):
 *   make_neighbor_pairwise_forcefield(apply_cutoff(softwell_potential<6>{energy, decay_distance}, cutoff))
 *       .set_neighbor_distance(cutoff).set_neighbor_targets(idx)
 * acts between every pair of TARGET beads closer than `cutoff`:
 *   U(r) = -energy / (1 + (r/decay)^6)  for r < cutoff, 0 beyond (plain truncation).
 * micromd's softwell_potential / apply_cutoff are not in the tree: sign and truncation are a DOCUMENTED
 * CHOICE.  Part of GD_TERM_PAIR; minimum image in periodic boxes.
 * At most 4096 targets; n_targets = 0 removes the term. */
int gd_set_pair_softwell(gd_system *sys, double energy, double decay, double cutoff,
                         const uint32_t *targets, uint32_t n_targets);

/* ------------------------------------------------------------- bonded (a6) */

enum {
    GD_POT_HARMONIC = 0,    /* U = K r^2 / 2                      md::harmonic_potential   */
    GD_POT_SPRING = 1,      /* U = K (r-b)^2 / 2                  md::spring_potential     */
    GD_POT_SEMISPRING = 2,  /* U = K (r-b)^2 / 2 for r>b else 0   md::semispring_potential */
    GD_POT_SOFTCORE = 3     /* U = k_a (1-(r/l_a)^p)^q, r<l_a     md::softcore_potential (glue_forcefield.cpp:12) */
};

typedef struct {
    int32_t kind;
    int32_t mix;                 /* 1: K=a*k_a+b*k_b, l=a*l_a+b*l_b with a,b the pair means
                                    (simulation_driver_forcefield.cc:60-70); 0: K=k_a, l=l_a */
    double  k_a, k_b, l_a, l_b;
    int32_t scale_by_bond_scale; /* K/=s^2, l*=s  (simulation_driver_forcefield.cc:72-77) */
    int32_t p, q;                /* GD_POT_SOFTCORE only */
    int32_t minimum_image;       /* periodic_box::shortest_displacement (glue_forcefield.cpp:38) */
} gd_bond_params;

/* add_bonded_range(start,end): bonds (i,i+stride) for start <= i, i+stride < end.
 * stride 1 = chain bonds (simulation_driver_forcefield.cc:86-88);
 * stride 2 = the (i,i+2) "loop" bonds (simulation_driver_forcefield.cc:123-127). */
int gd_add_bond_range(gd_system *sys, const gd_bond_params *p,
                      uint32_t start, uint32_t end, uint32_t stride);
/* add_bonded_pair(i,j) xn (simulation_driver_forcefield.cc:149-151). pairs = n x 2. */
int gd_add_bond_pairs(gd_system *sys, const gd_bond_params *p,
                      const uint32_t *pairs, uint32_t n);
/* Re-uploadable pair lists evaluated inside the step: the custom md::forcefield
 * subclasses loop_forcefield / glue_forcefield (3-sim-1kb/src/simulation/forces/
 * loop_forcefield.cpp:36-50, glue_forcefield.cpp:32-44). slot in [0,4). */
int gd_set_dynamic_pairs(gd_system *sys, uint32_t slot, const gd_bond_params *p,
                         const uint32_t *pairs, uint32_t n);

/* ------------------------------------------------------------ bending (a7) */

/* make_bonded_triplewise_forcefield(cosine_bending_potential).add_bonded_range(start,end):
 * triplets (i,i+1,i+2), U = e (1 - cos theta).  per_bead=0: e = energy
 * (simulation_spindle/simulation_driver.cc:119-130); per_bead=1: e = bending_energy of
 * the MIDDLE bead (3-sim-1kb/src/simulation/simulation.cpp:146-159). */
int gd_add_bending_range(gd_system *sys, uint32_t start, uint32_t end,
                         double energy, int per_bead);

/* ------------------------------------------------------- point source (a8) */

/* make_point_source_forcefield(pot).set_point_source(p)[.set_point_source_targets(idx)]
 * (simulation_spindle/simulation_driver.cc:147-155,163-171). kind: HARMONIC{K} or
 * SEMISPRING{K,b}. targets NULL => all beads. At most 4 sources. */
int gd_add_point_source(gd_system *sys, int kind, double k, double b,
                        const double point[3], const uint32_t *targets, uint32_t n_targets);

/* ------------------------------------------------- ellipsoid wall (a9,a11) */

typedef struct {
    /* inward: a_w*softcore<p_a,q_a>{eps_a, sigma_a/2 * bead_scale} + b_w*softcore<p_b,q_b>{...}
     * with a_w=(a_i+wall_a_factor)/2, b_w=(b_i+wall_b_factor)/2
     * (simulation_driver_forcefield.cc:196-214). sigma_* are FULL diameters here. */
    double  eps_a, sigma_a, eps_b, sigma_b;
    int32_t p_a, q_a, p_b, q_b;
    double  wall_a_factor, wall_b_factor;
    int32_t scale_by_bead_scale;
    /* outward: harmonic_potential{packing_spring} (simulation_driver_forcefield.cc:220-226) */
    double  packing_spring;
    /* wall ODE, simulation_driver_interphase.cc:70-80:
     * semiaxes += dt * mobility * (axial_reaction - semiaxes_spring (.) semiaxes) */
    double  semiaxes_spring[3];
    double  mobility;
    double  init_semiaxes[3];
} gd_wall;

int gd_set_ellipsoid_wall(gd_system *sys, const gd_wall *w);

/* Inner spherical wall around the origin, an excluded core
 * (4-sim-ab/sphere/src/simulation_driver.cc:184-228, active when inner_wall_radius >= 1e-6):
 *   beads outside it: make_sphere_outward_forcefield(a_w*softcore<p_a,q_a>{eps_a, sigma_a/2} + b_w*softcore<p_b,q_b>{eps_b, sigma_b/2})
 *                     on the displacement from the nearest surface point, a_w=(a_i+wall_a_factor)/2, b_w likewise;
 *   beads inside it:  make_sphere_inward_forcefield(harmonic_potential{spring}) pushing them back out.
 * sigma_* are FULL diameters; fold inner_wall_multiplier into eps_*.  Static (no dynamics, no reaction).
 * Part of the GD_TERM_WALL term; independent of gd_set_ellipsoid_wall. */
typedef struct {
    double  radius;
    double  eps_a, sigma_a, eps_b, sigma_b;
    int32_t p_a, q_a, p_b, q_b;
    double  wall_a_factor, wall_b_factor;
    double  spring;
} gd_inner_sphere;

int gd_set_inner_sphere_wall(gd_system *sys, const gd_inner_sphere *w);

/* --------------------------------------------------- per-step context (a11) */

/* bead_scale(t) = 1-(1-init)exp(-t/tau), same for bond_scale
 * (simulation_driver_interphase.cc:59-67). Initial context = {init, init}
 * (simulation_interphase/simulation_driver.cc:43-51). */
int gd_set_scaling(gd_system *sys, double bead_scale_init, double bead_scale_tau,
                   double bond_scale_init, double bond_scale_tau);

typedef struct {
    int64_t  step;           /* steps taken in the current phase */
    double   time;           /* step * timestep of the last run */
    double   bead_scale, bond_scale;
    double   semiaxes[3];
    double   axial_reaction[3];   /* stats.axial_reaction of the last force evaluation */
    /* diagnostics (not part of the reference's state) */
    uint64_t list_entries;   /* L: directed neighbour-list entries currently stored */
    uint64_t rebuilds;       /* neighbour-list builds since creation */
    uint64_t rollbacks;      /* verified-skin rollbacks since creation */
    uint32_t rebuild_interval;
    double   list_radius;
    uint32_t list_path;      /* kernel path of the list in use: 0 none yet, 1 generic (global gather), 2 LDS-tiled */
    uint32_t callback_pending;   /* 1: the state updates of callback(step + 1) are pending (GD_RUN_DEFER_CALLBACK) */
    uint32_t tile_capacity;      /* LDS-tiled lists: beads of LDS per block the list in use was built for (0 on the generic path) */
    uint32_t compensated;        /* 1: the last gd_run stepped with the compensated position update (gd_run, below) */
    uint32_t largest_tile;       /* LDS-tiled lists: beads in the largest tile of the last build, over all replicas (what decides the
                                    tile class, and with it the list width the handle selects) */
    uint32_t row_repairs;        /* LDS-tiled lists: k_step waves whose rows the builds of the last chunk wrote twice (a list outgrew the
                                    width predicted for its wave: repaired on the device, no rollback) */
    uint64_t near_entries;       /* LDS-tiled lists: entries of this replica's NEAR class, in the fours k_step walks them in (every step
                                    walks these; the rest of list_entries only once displacements make it matter) */
    uint64_t list_bytes;         /* memory the lists of the whole handle occupy: tiled lists the rows taken from the pool by the last
                                    build (ragged rows: every wave as wide as its longest list), generic lists uniform rows */
} gd_context;

int gd_get_context(gd_system *sys, uint32_t replica, gd_context *out);
/* Start a phase: step=0, time=0, semiaxes as given (NULL keeps), scales re-evaluated at t=0.
 * Mirrors the drivers' setup_context() + the driver-called callback(0). */
int gd_begin_phase(gd_system *sys, const double *semiaxes /* (R,3) or NULL */);
/* Restart support (simulation_fine_sampling/simulation_driver.cc:44-54). */
int gd_set_context(gd_system *sys, uint32_t replica, int64_t step,
                   double bead_scale, double bond_scale, const double semiaxes[3]);

/* ---------------------------------------------------------- stepping (a1) */

enum { GD_NOISE_PHILOX = 0, GD_NOISE_ZERO = 1, GD_NOISE_HOST = 2 };
enum {
    GD_RUN_UPDATE_SCALES = 1,   /* callback runs update_bead_scale()   (simulation_driver_interphase.cc:42) */
    GD_RUN_WALL_DYNAMICS = 2,   /* callback runs update_wall_semiaxes() (simulation_driver_interphase.cc:43) */
    GD_RUN_COMPENSATED = 8,     /* force the compensated position update (below) */
    GD_RUN_UNCOMPENSATED = 16,  /* never use it */
    GD_RUN_DEFER_CALLBACK = 4   /* the state updates of the LAST step's callback are left pending when gd_run returns: positions are
                                   those of step k, the context is the one callback(k-1) left -- what the reference's callback(k)
                                   sees when it computes mean_energy, prints and saves (simulation_driver_interphase.cc:20-22,
                                   29-32) BEFORE update_bead_scale() / update_wall_semiaxes() (:42-43).  gd_compute_energy,
                                   gd_get_context, gd_get_positions*, gd_search_pairs observe that state; the pending updates are
                                   applied by gd_apply_callback(), by the next gd_run() or by gd_compute_forces() */
};

/* md::simulate_brownian_dynamics(system, {temperature,timestep,spacestep,steps,seed,callback})
 * (simulation_driver_interphase.cc:48-55): for k=1..steps
 *     x_i += mu_i F_i dt + sqrt(2 mu_i kT dt) xi_i ;  callback(k)
 * The callback's per-step state updates run on the device (flags); everything else a
 * callback does (logging, sampling) is done by the caller between gd_run() chunks.
 *
 * Compensated positions.  The reference integrates in fp64; the device keeps fp32 coordinates.  When the increment of a
 * step comes within a few dozen ulp of a coordinate -- the deterministic fine-sampling run, T = 0 and dt = 1e-7
 * (simulation_fine_sampling/simulation_driver.cc:30-34) -- the run keeps an fp32 residual per coordinate and adds
 * increments by a two-sum, so that a bead's position is pos + residual (about 48 significant bits) and no part of
 * mu F dt is lost; gd_get_positions returns the sum, gd_get_positions_f32 its fp32 part (float(pos + residual), which is what
 * simulation_store.cc:403-407 quantises).  Selected by gd_run when
 * sqrt(2 mu_max kT dt) < 64 ulp(largest wall semiaxis / box period; 16 without either), i.e. always at T = 0;
 * GD_RUN_COMPENSATED / GD_RUN_UNCOMPENSATED override.  gd_set_positions seeds the residuals with what fp32 drops of the
 * fp64 input; a run without compensation discards them. */
typedef struct {
    double   temperature;
    double   timestep;
    double   spacestep;     /* must be 0 (all reference defaults, config_entries.inc:61,71,79) */
    int64_t  steps;
    uint64_t seed;          /* replica r draws from stream (seed, r); see replica_seeds */
    int32_t  noise_mode;
    int32_t  flags;
    const double *host_noise;   /* GD_NOISE_HOST: (steps, R, N, 3) standard normals */
    const uint64_t *replica_seeds;  /* optional (R): replica r draws from stream (replica_seeds[r], 0) -- the stream a
                                       one-replica run with seed replica_seeds[r] uses -- instead of (seed, r); this is
                                       how R runs of the reference's one-process-per-seed ensemble
                                       (5-sim-genome/scripts/run_simulation:8-25) are batched into one handle */
} gd_run_desc;

int gd_run(gd_system *sys, const gd_run_desc *run);
/* Applies the state updates a GD_RUN_DEFER_CALLBACK run left pending (update_bead_scale / update_wall_semiaxes of its last
 * callback, simulation_driver_interphase.cc:42-43) with that run's timestep and flags.  No-op when nothing is pending. */
int gd_apply_callback(gd_system *sys);

/* ------------------------------------------------------------- observation */

enum {
    GD_TERM_PAIR = 1, GD_TERM_BOND = 2, GD_TERM_BEND = 4, GD_TERM_POINT = 8,
    GD_TERM_WALL = 16, GD_TERM_DYNAMIC = 32, GD_TERM_ALL = 63
};

/* md::system::compute_energy() (simulation_driver_interphase.cc:20-22): per replica. */
int gd_compute_energy(gd_system *sys, uint32_t term_mask, double *energy /* (R) */);
/* Forces on the current positions under the current context, (R,N,3) fp64;
 * also refreshes axial_reaction. Parity/diagnostic entry point. */
int gd_compute_forces(gd_system *sys, uint32_t term_mask, double *forces);

/* md::neighbor_searcher<Box>{box,dcut}.set_points().search(out)
 * (simulation_interphase/contact_map.cc:64-66, 3-sim-1kb/src/simulation/glues/glue_simulator.cpp:41,67-77): unique pairs
 * i<j within dcut of replica `replica`, in no particular order; writes up to cap pairs, returns the total in *n_pairs.
 * Served on the device from the Verlet list that is resident (no rebuild while it covers dcut; otherwise one build at
 * max(force-list radius, dcut), which leaves a valid force list behind).  Calling it twice without an intervening change of
 * the positions (count, then fetch) costs one search. */
int gd_search_pairs(gd_system *sys, uint32_t replica, double dcut,
                    uint32_t *pairs, uint64_t cap, uint64_t *n_pairs);

/* contact_map (5-sim-genome/src/simulation_interphase/contact_map.hpp, contact_map.cc:26-91): the time-integrated contact map
 * of EVERY replica of the handle, kept on the device between two dumps.
 *   gd_contacts_update(distance)   contact_map::update(points) (contact_map.cc:31-74; call site
 *                                  simulation_driver_interphase.cc:33-35) for all replicas at once: every unique pair i<j within
 *                                  `distance` (the handle's box decides the metric, as for gd_search_pairs) is counted once more.
 *                                  One pair search over all replicas + one insert launch; the pairs never leave the device.
 *   gd_contacts_fetch(replica,...) contact_map::accumulate() (:77-91; simulation_driver_interphase.cc:37-38): rows (i, j, count),
 *                                  count > 0, in row-major order (i ascending, then j); writes up to cap rows of three
 *                                  uint32, returns the total in *n_rows (rows == NULL, cap == 0: the count only, free).
 *   gd_contacts_clear(replica)     contact_map::clear() (:26-29) of one replica, or of all (GD_ALL_REPLICAS).
 * The map is keyed by bead index and survives gd_set_positions / gd_begin_phase; gd_destroy frees it. */
#define GD_ALL_REPLICAS 0xffffffffu
int gd_contacts_update(gd_system *sys, double distance);
int gd_contacts_fetch(gd_system *sys, uint32_t replica, uint32_t *rows, uint64_t cap, uint64_t *n_rows);
int gd_contacts_clear(gd_system *sys, uint32_t replica);

/* --------------------------------------------------------- tuning / timing */

typedef struct {
    double   skin;              /* Verlet skin as a fraction of the NOMINAL pair cutoff: list radius = cutoff * (bead_scale + skin),
                                   i.e. an absolute width that a scaled-down cutoff does not shrink.  A value > 0 pins the width.
                                   0 keeps the library's own choice: 0.75, widened to 0.9 while the largest LDS tile of the wider
                                   list fits the three-block class (a rule on the state: tile sizes are cell counts; see
                                   DESIGN.md section 4), narrowed while a dense state's largest tile would not fit the LDS (or its
                                   rows a sixteenth of the device memory); < 0 returns a pinned handle to the library's choice */
    uint32_t rebuild_interval;  /* initial steps between list builds; 0 = auto */
    uint32_t adapt_interval;    /* 1 = adapt interval from measured displacements */
    uint32_t list_width;        /* neighbours per bead the first list build allows for: the uniform row width of generic lists (grows on
                                   overflow), the guess the ragged rows of tiled lists start from before a build has counted the needs */
    uint32_t kernel_path;       /* 0 auto (LDS-tiled where the tiles fit, else generic), 1 generic (global-gather lists),
                                   2 LDS-tiled preferred: needs fp16-exact a/b factors, tiles that fit the LDS and rows of at most
                                   1016 near + 504 far entries; where that does not hold the build falls back to generic lists (a dense
                                   transient) and retries later -- gd_context.list_path reports the path of the list in use */
    double   near_fraction;     /* tiled lists keep the entries closer than cutoff + near_fraction x skin width in a "near" class
                                   that every step walks, the rest in a "far" class that is walked only once displacements
                                   make it matter (results do not depend on it beyond summation order); 0 keeps the current
                                   value (default 0.65); a value near 0 makes every step walk both classes */
    uint32_t auto_skin;         /* 1: select the skin for this workload from measured chunk times (a few candidate widths, each run
                                   for a few verified chunks once the rebuild interval has settled; repeated when the interval
                                   drifts).  Costs a few thousand steps at candidate widths; results are independent of the skin
                                   (verified lists), so only the cost changes.  0 (default): keep `skin` */
} gd_tuning;

int gd_set_tuning(gd_system *sys, const gd_tuning *t);

typedef struct {
    double   step_kernel_ms;    /* HIP-event time of the step kernels in the last gd_run */
    double   rebuild_ms;        /* HIP-event time of the list builds in the last gd_run  */
    double   total_ms;          /* HIP-event time of the whole enqueued region           */
    uint64_t step_launches, rebuild_launches;
    uint64_t list_entries_visited;  /* sum over step launches of L, the directed entries stored (all replicas; tiled lists skip
                                       the far class in most steps, so fewer are evaluated) */
} gd_timing;

int gd_get_timing(gd_system *sys, gd_timing *out);
/* Stream the handle enqueues on (hipStream_t as void*), for callers' own events. */
int gd_get_stream(gd_system *sys, void **stream);

#ifdef __cplusplus
}
#endif
#endif /* GDYN_H */
