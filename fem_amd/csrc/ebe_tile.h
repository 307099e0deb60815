// fem_amd/csrc/ebe_tile.h -- the tile plan of the one-pass element-by-element product (ebe_tile.hip: kernels and the host
// builder kept for cross-checks; ebe_plan.hip: the builder that runs on the device).
#pragma once
#include <memory>

#include "objects.h"

namespace fem {

constexpr int TILE_MAXC = 16;            // colours a tile may need (more: the mesh keeps the two-kernel product)
constexpr int TILE_GEO_STRIDE = 48;   // doubles per tile in EbeTilePlan::pgeo (hex8)
constexpr int TILE_X_MAX_NODES = 1536, TILE_X_MAX_OWN = 768;   // tiles of the kernels with x in LDS (k_ebe_tile_x): 6 / 3 slots per thread
constexpr int TILE_X_LDS_NODES = 1408;   // their LDS image, fixed (the component offsets become immediates of the LDS instructions):
                                         // (1408 + 768) * 24 B = 51 KiB, three workgroups per CU; the 16 x 8 x 8-element tile has 1,377 nodes
constexpr int TILE_MAX_NODES = 2900;     // nodes of a tile: 28 B of LDS each, two workgroups per CU (160 KiB); <= 12 * 256

struct EbeTilePlan {
    int ntiles = 0, n_aff = 0, n_gen = 0, max_nnode = 0, max_nsteps = 0, max_nown = 0;
    long long tile_elements = 0;         // elements computed per product (rim elements counted once per tile)
    long long slots = 0;                 // lanes of all steps (elements + the padding of 16-lane rows)
    long long tile_nodes = 0;            // entries of all node lists
    DBuf<int> hdr;                       // [ntiles*6] node_off, nown, nnode, elem_off, step_off, nsteps
    DBuf<int2> steps;                    // rounds of <= 256 elements of one colour: {first (relative), count | barrier << 31}
    DBuf<int> nodes;                     // global node ids of every tile, owned ones first
    DBuf<int> elems;                     // [slots] global element id of every lane slot, colour by colour; -1 = padding
    DBuf<uint4> conn16;                  // the same elements' eight tile-local node numbers (16 bit each)
    DBuf<unsigned char> flags;           // [slots] bit 0: the element's +x face is the -x face of the next lane's element
    DBuf<int> list_aff, list_gen;        // tiles whose elements are all parallelepipeds / the others; in both lists the tiles
                                         // that own a node shared with another domain come first
    int n_aff_if = 0, n_gen_if = 0;      // how many of each list those are
    DBuf<int> list_cart;                 // tiles of axis-aligned bricks (a third list, same order rule); not in list_aff
    DBuf<int> lhdr_cart, lhdr_aff, lhdr_gen;   // [n * 8] the listed tiles' hdr records + tile id, in list order (what the kernels read)
    // the listed tiles' node lists once more at a FIXED stride per list position (the kernels with x in LDS): the load of a tile's node
    // ids then needs nothing from its record -- header and ids travel together instead of one behind the other (one round trip less at the
    // start of every tile)
    DBuf<int> pnodes_cart, pnodes_aff, pnodes_gen;
    int pnode_stride = 0;
    // ... and, for the same reason, by global list position gp (cart tiles, then aff, then gen): the tile's step table (64 entries, zeros
    // behind the last step), the slot records of its FIRST step (256 lanes; zeros where the step has no element) and, where the tile
    // has one geometry record for all its elements (hex8, lhdr[7]) or per lane slot (quad4: the two D factors of the first step), that
    // too.  With these the only load of a tile's start that waits for another one is the gather of x behind the node ids.
    DBuf<int2> psteps;                   // [npos * 64]
    int pslot_steps = 1;                 // steps whose slot records are stored that way: 4 where no tile has more (then ALL of a tile's
                                         // slot records leave with its record), else the first
    DBuf<uint4> pconn0;                  // [npos * pslot_steps * 256]
    DBuf<unsigned char> pfl0;            // [npos * pslot_steps * 256]
    DBuf<double> pgeo;                   // hex8: [npos * TILE_GEO_STRIDE] the tile's record (valid where lhdr[7] > 0; entries 14, 15: E and nu of its elements; from 16: the 30 coefficients of hex_forces_cart_k for a tile of bricks); quad4: [npos * pslot_steps * 512] lam, mu of those steps
    int n_cart = 0, n_cart_if = 0;
    const void* halo_seen = nullptr;     // the mesh's Halo when the plan was built (the split above is only valid for that one)
    DBuf<double> geo;                    // [11][tile_elements] stored geometry of the affine tiles' elements (hex_affine_geom)
    double t_setup_ms = 0.0;
    long long n_uniform = 0;   // tiles of list_cart / list_aff whose elements share one geometry record (read once per tile: lhdr[7])
    bool stream = false;   // lists + the CG's vectors beyond the caches: the product loads the lists with the streaming hint (ebe_tile.hip)
};


// Grid coordinate of a node (in spacings h from the origin lo), rounded to the nearest grid point.  The millionth of a spacing
// added before the rounding settles the nodes that sit half-way between two grid points (a graded mesh measured in its mean
// spacing has whole planes of them): without it the last bit of lo or h -- sums whose order differs between the builders --
// would decide their box.
#define FEM_PLAN_GRID(c, lo, h) llround(((c) - (lo)) / (h) + 1.0e-6)

// limits of the builders (both build the same plan): elements of one tile (a set of owned nodes that touches more is halved)
constexpr int PLAN_MAX_ELEMS = 2048;

}  // namespace fem

std::shared_ptr<fem::EbeTilePlan> ebe_tile_build(fem_mesh m, hipStream_t s);          // dispatcher (ebe_tile.hip)
std::shared_ptr<fem::EbeTilePlan> ebe_tile_build_host(fem_mesh m, hipStream_t s);     // host worker threads (cross-check: FEM_AMD_TILE_PLAN=host)
std::shared_ptr<fem::EbeTilePlan> ebe_tile_build_device(fem_mesh m, hipStream_t s);   // ebe_plan.hip
// geometry records of the stored-geometry tiles' elements (taff: class per lane slot), ebe_tile.hip
void ebe_tile_list_headers(fem::EbeTilePlan& P, bool quad, hipStream_t s);
void ebe_tile_first_records(fem_mesh m, fem::EbeTilePlan& P, bool quad, hipStream_t s);   // psteps, pconn0, pfl0, pgeo (after the stored geometry)   // lhdr_* from list_* and hdr (ebe_tile.hip)
void ebe_tile_quad_factors(fem_mesh m, fem::EbeTilePlan& P, hipStream_t s);   // quad4 plans: D factors per lane slot
void ebe_tile_store_geometry(fem_mesh m, fem::EbeTilePlan& P, const unsigned char* d_taff, hipStream_t s);

