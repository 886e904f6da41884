// seg_solve60.hip -- the instances of k_solve (seg_solve.h) for the size class of problems with up to 60 candidates.
// Part of libfreddie_seg.so (gfx950).
#include "seg_solve.h"

namespace fseg {

__attribute__((used)) static const void *const kInstances[] = {
    reinterpret_cast<const void *>(&k_solve<kNMax, unsigned char, int, false>),
    reinterpret_cast<const void *>(&k_solve<kNMax, unsigned char, i64, false>),
    reinterpret_cast<const void *>(&k_solve<kNMax, unsigned short, int, false>),
    reinterpret_cast<const void *>(&k_solve<kNMax, unsigned short, i64, false>),
    reinterpret_cast<const void *>(&k_solve<kNMax, unsigned char, int, true>),
    reinterpret_cast<const void *>(&k_solve<kNMax, unsigned short, int, true>),
};

}  // namespace fseg
