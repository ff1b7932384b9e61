// condensed.hpp -- tables and host entry points of the condensed (face-only) assembly, condensed.hip.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "structured_mesh.hpp"

namespace pa {

// what the condensed assembly kernels know of the mesh: the assembler's face tables (hho_assembly.hpp) plus the two
// cells of every local face (adj[2f], adj[2f+1]: lower cell id first; 0x7fffffff / -1 = none, equal = one cell)
struct CondMesh {
    const uint32_t *cell_faces;
    const int32_t *face_compress;
    const int32_t *adj;
    StructuredMesh sm;
    bool structured;
};

// symbolic record of an owned non-Dirichlet face (position = its compressed id - the first owned one)
struct CondFace {
    int32_t cA, cB;          // its cells (local index), -1 none; cA <= -2: cell -2 - cA of the slab below (remote)
    int32_t colcomp[7];      // compressed ids of its column faces, ascending; -1 padded
    uint8_t code[7];         // per column face: bits 0-1 local face index in cell A, bit 2 present in A; bits 3-4, 5: cell B
    uint8_t rows;            // bits 0-1 / 2-3: the face's own local index in cell A / B
    uint8_t ncol;
    uint32_t face;           // local face index
};

// the part of it the numeric phase reads, one 16-byte load: bits 6s .. 6s+5 of `packed`: code[s]; bits 42-45: rows;
// bits 46-48: ncol; bit 49: some face of its local cells is Dirichlet (their data enter the right-hand side)
struct alignas(16) CondFaceLean {
    int32_t cA, cB;
    uint64_t packed;
};

hipError_t cond_build_tables(hipStream_t stream, CondMesh m, uint32_t nfaces_local, uint32_t ncells, uint32_t owned_range,
                             int32_t p0, uint32_t nown, int32_t *adj, CondFace *faces, CondFaceLean *lean, uint32_t *ncols,
                             uint32_t *prefix);
hipError_t cond_pattern(hipStream_t stream, uint32_t nown, int fbs, const CondFace *faces, const uint32_t *prefix, int64_t *rowptr,
                        int32_t *colind);
hipError_t cond_fill(hipStream_t stream, const CondMesh &m, uint32_t nown, int fbs, const CondFaceLean *lean, const uint32_t *prefix,
                     const double *cond, const double *g, const double *halo, double *values, double *rhs);
hipError_t cond_halo_pack(hipStream_t stream, const CondMesh &m, uint32_t first_cell, uint32_t ncells_row, int fbs, const double *cond,
                          const double *g, double *halo);
hipError_t cond_triplets(hipStream_t stream, const CondMesh &m, int num_cus, size_t first, size_t n, int fbs, const double *cond,
                         const double *g, int32_t *rows, int32_t *cols, double *vals, int32_t *rhs_rows, double *rhs_vals);
hipError_t cond_take_faces(hipStream_t stream, const CondMesh &m, size_t first, size_t n, int fbs, const double *solution,
                           const double *g, double *uF);
hipError_t cond_expand(hipStream_t stream, size_t ncells_local, size_t cell_base, size_t ncells_global, int cbs, size_t nface_dofs,
                       const double *uT, const double *xF, double *full);

}  // namespace pa
