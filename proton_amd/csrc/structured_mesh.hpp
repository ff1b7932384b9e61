// structured_mesh.hpp -- closed forms of the generator mesh mesh_impl<T,4>(mesh_init_params)
// (src/core/core_bits/basic_mesh.hpp:230-298): global face ids, Dirichlet flags and the assembler's compress table
// (src/methods/hho_bits/hho.hpp:305-323) of a slab of cell rows.  Host and device.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pa {

struct StructuredMesh {
    uint32_t Nx, Ny, row0, row1;      // the context owns cell rows [row0, row1)
};

__host__ __device__ inline uint32_t sm_face_row(const StructuredMesh &m) { return 2 * m.Nx + 1; }
// global ids of the faces of the generator mesh
__host__ __device__ inline uint32_t sm_hface(const StructuredMesh &m, uint32_t i, uint32_t j)
{
    return j < m.Ny ? j * sm_face_row(m) + 2 * i : m.Ny * sm_face_row(m) + i;
}
__host__ __device__ inline uint32_t sm_vface(const StructuredMesh &m, uint32_t i, uint32_t j)
{
    return j * sm_face_row(m) + (i < m.Nx ? 2 * i + 1 : 2 * m.Nx);
}
__host__ __device__ inline uint32_t sm_face_base(const StructuredMesh &m) { return m.row0 * sm_face_row(m); }
__host__ __device__ inline uint32_t sm_faces_local(const StructuredMesh &m)
{
    // rows row0..row1-1 in full, plus the horizontals that close the slab on top (they live in
    // row row1's block, or in the top row)
    return (m.row1 - m.row0) * sm_face_row(m) + (m.row1 < m.Ny ? sm_face_row(m) : m.Nx);
}
__host__ __device__ inline uint32_t sm_num_other_faces(const StructuredMesh &m)
{
    return m.Nx * (m.Ny + 1) + m.Ny * (m.Nx + 1) - 2 * (m.Nx + m.Ny);
}

// decode a global face id: endpoints (global point ids, lo < hi), Dirichlet flag (every boundary
// face, basic_mesh.hpp:293-297) and the compress-table value (hho.hpp:313-323) in closed form
__host__ __device__ inline void sm_face_decode(const StructuredMesh &m, uint32_t gid, uint32_t &lo, uint32_t &hi,
                                               bool &dirichlet, int32_t &compress)
{
    const uint32_t row = sm_face_row(m), npr = m.Nx + 1;
    if (gid >= m.Ny * row) {                         // top row: horizontals only, all on the boundary
        const uint32_t i = gid - m.Ny * row;
        lo = m.Ny * npr + i; hi = lo + 1; dirichlet = true; compress = -1;
        return;
    }
    const uint32_t jj = gid / row, pos = gid % row;
    // non-Dirichlet faces in the rows below: row 0 has Nx-1 interior verticals, every other row
    // Nx horizontals and Nx-1 interior verticals
    uint32_t cnt = jj >= 1 ? (m.Nx - 1) + (jj - 1) * (2 * m.Nx - 1) : 0;
    // ... and in this row at positions < pos: horizontals sit at even positions 2i, verticals at
    // 2i+1 (i < Nx) and at 2Nx (i = Nx)
    const uint32_t nh = (pos + 1) / 2 < m.Nx ? (pos + 1) / 2 : m.Nx;
    if (jj > 0) cnt += nh;
    const uint32_t nv_all = pos / 2;                 // verticals i = 0 .. nv_all-1 lie before pos
    uint32_t nv_int = nv_all > 0 ? nv_all - 1 : 0;   // i = 0 is on the boundary
    if (nv_int > m.Nx - 1) nv_int = m.Nx - 1;
    cnt += nv_int;
    const bool horizontal = (pos % 2 == 0) && pos < 2 * m.Nx;
    if (horizontal) {
        const uint32_t i = pos / 2;
        lo = jj * npr + i; hi = lo + 1;
        dirichlet = jj == 0;
    } else {
        const uint32_t i = pos == 2 * m.Nx ? m.Nx : pos / 2;
        lo = jj * npr + i; hi = lo + npr;
        dirichlet = (i == 0) || (i == m.Nx);
    }
    compress = dirichlet ? -1 : (int32_t)cnt;
}

}  // namespace pa
