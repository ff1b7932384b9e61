// assembler_csr.hpp -- host entry points of assembler_csr.hip: the reference's own global system (cell + face unknowns,
// hho.hpp:252-463) in CSR, built from the face adjacency tables of condensed.hpp.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "condensed.hpp"

namespace pa {

// per cell the number of its non-Dirichlet faces (nfc, ncells + 1) and its exclusive prefix; per non-Dirichlet face the number of
// its cells (nfcell, nown + 1) and its exclusive prefix
hipError_t asm_build_tables(hipStream_t stream, const CondMesh &m, uint32_t ncells, uint32_t nown, const CondFaceLean *lean,
                            uint32_t *nfc, uint32_t *cprefix, uint32_t *nfcell, uint32_t *fprefix);
hipError_t asm_pattern(hipStream_t stream, const CondMesh &m, int cbs, int fbs, uint32_t ncells, uint32_t nown, uint64_t cell_nnz,
                       const CondFace *faces, const uint32_t *colprefix, const uint32_t *cprefix, const uint32_t *fprefix,
                       int64_t *rowptr, int32_t *colind);
hipError_t asm_fill(hipStream_t stream, const CondMesh &m, int cbs, int fbs, uint32_t ncells, uint32_t nown, uint64_t cell_nnz,
                    const CondFaceLean *lean, const uint32_t *colprefix, const uint32_t *cprefix, const uint32_t *fprefix,
                    const double *lc, const double *rhs, const double *g, double *values, double *RHS);

}  // namespace pa
