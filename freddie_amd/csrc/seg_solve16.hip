// seg_solve16.hip -- the instances of k_solve (seg_solve.h) for the size class of problems with up to 16 candidates.
// Part of libfreddie_seg.so (gfx950).
#include "seg_solve.h"

namespace fseg {

__attribute__((used)) static const void *const kInstances[] = {
    reinterpret_cast<const void *>(&k_solve<kClsSmall, unsigned char, int, false>),
    reinterpret_cast<const void *>(&k_solve<kClsSmall, unsigned char, i64, false>),
    reinterpret_cast<const void *>(&k_solve<kClsSmall, unsigned short, int, false>),
    reinterpret_cast<const void *>(&k_solve<kClsSmall, unsigned short, i64, false>),
    reinterpret_cast<const void *>(&k_solve<kClsSmall, unsigned char, int, true>),
    reinterpret_cast<const void *>(&k_solve<kClsSmall, unsigned short, int, true>),
};

}  // namespace fseg
