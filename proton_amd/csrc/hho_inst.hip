// hho_inst.hip -- one translation unit per (cell degree, face degree, quadrature kind):
// compiled with -DPA_CD= -DPA_FD= -DPA_QUAD= -DPA_GMIN= (see pa_configs.def, _build.py).
#include "hho_launch.hpp"

#if !defined(PA_CD) || !defined(PA_FD) || !defined(PA_QUAD) || !defined(PA_GMIN)
#error "compile with -DPA_CD -DPA_FD -DPA_QUAD -DPA_GMIN"
#endif

#define PA_STR2(x) #x
#define PA_STR(x) PA_STR2(x)
#define PA_CAT5(a, b, c, d, e) pa_entries_##a##_##b##_##c
#define PA_COND_NONE nullptr, nullptr, 0, nullptr, 0
// (the condensed mode needs a lane per row of [lc f_T; f_T^T 0]: msize + 1 <= G)
#define PA_COND_OF(STAB, G)                                                                         \
    (pa::Cfg<PA_CD, PA_FD, PA_QUAD, STAB, G>::MS + 1 <= G)                                          \
        ? &pa::launch_local_ops<pa::Cfg<PA_CD, PA_FD, PA_QUAD, STAB, (pa::Cfg<PA_CD, PA_FD, PA_QUAD, STAB, G>::MS + 1 <= G ? G : 64), 1>, pa::MODE_COND> \
        : nullptr,                                                                                  \
    (const void *)&pa::hho_local_ops_kernel<pa::Cfg<PA_CD, PA_FD, PA_QUAD, STAB, (pa::Cfg<PA_CD, PA_FD, PA_QUAD, STAB, G>::MS + 1 <= G ? G : 64), 1>, pa::MODE_COND>, \
    (int)(pa::Cfg<PA_CD, PA_FD, PA_QUAD, STAB, (pa::Cfg<PA_CD, PA_FD, PA_QUAD, STAB, G>::MS + 1 <= G ? G : 64), 1>::LDS_DOUBLES * sizeof(double)), \
    "hho_condensed_ops<cd=" PA_STR(PA_CD) ",fd=" PA_STR(PA_FD) ",quad=" PA_STR(PA_QUAD) ",stab=" #STAB ",G=" #G ">", \
    pa::Cfg<PA_CD, PA_FD, PA_QUAD, STAB, (pa::Cfg<PA_CD, PA_FD, PA_QUAD, STAB, G>::MS + 1 <= G ? G : 64), 1>::WAVES
// the thread-per-cell kernel of the small pairs (msize <= 9: (0,0), (1,0), (0,1)); instantiated for those only
#if (PA_CD + 2) * (PA_CD + 1) / 2 + 4 * (PA_FD + 1) <= 9
#define PA_SMALL_OF(STAB, G) &pa::launch_small_ops<pa::Cfg<PA_CD, PA_FD, PA_QUAD, STAB, G>>
#else
#define PA_SMALL_OF(STAB, G) nullptr
#endif
#define PA_ENTRY(STAB, G, COND)                                                                    \
    {PA_CD, PA_FD, PA_QUAD, STAB, G, &pa::launch_local_ops<pa::Cfg<PA_CD, PA_FD, PA_QUAD, STAB, G>, pa::MODE_LC>, \
     &pa::launch_local_ops<pa::Cfg<PA_CD, PA_FD, PA_QUAD, STAB, G>, pa::MODE_SPLIT>,                 \
     (const void *)&pa::hho_local_ops_kernel<pa::Cfg<PA_CD, PA_FD, PA_QUAD, STAB, G>, pa::MODE_LC>,  \
     (int)(pa::Cfg<PA_CD, PA_FD, PA_QUAD, STAB, G>::LDS_DOUBLES * sizeof(double)),                  \
     "hho_local_ops<cd=" PA_STR(PA_CD) ",fd=" PA_STR(PA_FD) ",quad=" PA_STR(PA_QUAD) ",stab=" #STAB ",G=" #G ">",                \
     pa::Cfg<PA_CD, PA_FD, PA_QUAD, STAB, G>::WAVES,                                                \
     pa::Cfg<PA_CD, PA_FD, PA_QUAD, STAB, G>::USE_PRE ? &pa::launch_pre<pa::Cfg<PA_CD, PA_FD, PA_QUAD, STAB, G>> : nullptr, \
     pa::Cfg<PA_CD, PA_FD, PA_QUAD, STAB, G>::Pre::NPRE,                                            \
     PA_SMALL_OF(STAB, G), "hho_small_ops<cd=" PA_STR(PA_CD) ",fd=" PA_STR(PA_FD) ",quad=" PA_STR(PA_QUAD) ",stab=" #STAB ">", \
     pa::Cfg<PA_CD, PA_FD, PA_QUAD, STAB, G>::SELF_PRE ? 1 : 0, COND}
#define PA_ENTRIES_G(G) PA_ENTRY(0, G, PA_COND_NONE), PA_ENTRY(1, G, PA_COND_OF(1, G)), PA_ENTRY(2, G, PA_COND_OF(2, G))

static const pa::KernelEntry k_entries[] = {
#if PA_GMIN <= 16
    PA_ENTRIES_G(16),
    PA_ENTRIES_G(32)
#else
    PA_ENTRIES_G(32),
    PA_ENTRIES_G(64)
#endif
};

#define PA_FN2(cd, fd, q) pa_entries_##cd##_##fd##_##q
#define PA_FN(cd, fd, q) PA_FN2(cd, fd, q)
// internal to the library (the registry of capi.hip): not part of the exported C ABI
extern "C" __attribute__((visibility("hidden"))) const pa::KernelEntry *PA_FN(PA_CD, PA_FD, PA_QUAD)(int *count)
{
    *count = (int)(sizeof(k_entries) / sizeof(k_entries[0]));
    return k_entries;
}
