// host_pack.h -- the atoms a HOST call of the distance functions selects, packed before the upload (round 6).
// The reference's drivers hand every call the whole trajectory (projections/util.py:40-70: mol.coords) and selections that are
// usually a few hundred atoms of a solvated system: MetricDistance's protein C-alpha x ligand call reads 330 of 30 000 atoms --
// 8 MB of a 737-MB array -- and the host entry points uploaded all of it, 15 ms in front of a 36-us kernel.  The rows of an atom
// are contiguous in the reference's layout ([atom][3][frame]), so packing is one memcpy per selected atom; selections, group atom
// lists, chain ids and masses are rewritten to the packed numbering, and what comes back as atom indices (contact lists) is
// translated back.  Plain C++ (no HIP): compiled into the library's host entry points and, for the tests, into the emulator.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

namespace mkamd {

struct PackedAtoms {
    bool on = false;                     // false: the call uploads the array as it is
    std::vector<uint32_t> uniq;          // packed index -> atom (ascending)

    template <class T>
    void collect(const T* idx, int64_t n) { for (int64_t i = 0; i < n; ++i) uniq.push_back((uint32_t)idx[i]); }

    // Decide, and pack the coordinates into `out` ([M, 3, F], grown as needed): worth it when the selected atoms are at most a quarter
    // of the array's and the array is more than a megabyte
    bool finish(const float* coords, int64_t N, int64_t F, std::vector<float>& out)
    {
        std::sort(uniq.begin(), uniq.end());
        uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
        const size_t M = uniq.size(), row = (size_t)3 * (size_t)F;
        on = M > 0 && M * 4 <= (size_t)N && (size_t)N * row * 4 > ((size_t)1 << 20);
        if (!on) return false;
        if (out.size() < M * row) out.resize(M * row);
        for (size_t k = 0; k < M; ++k) std::memcpy(out.data() + k * row, coords + (size_t)uniq[k] * row, row * sizeof(float));
        return true;
    }
    int64_t size() const { return (int64_t)uniq.size(); }
    uint32_t packed(uint32_t atom) const { return (uint32_t)(std::lower_bound(uniq.begin(), uniq.end(), atom) - uniq.begin()); }
    template <class T>
    std::vector<T> remap(const T* idx, int64_t n) const
    {
        std::vector<T> r((size_t)n);
        for (int64_t i = 0; i < n; ++i) r[(size_t)i] = (T)packed((uint32_t)idx[i]);
        return r;
    }
    template <class T>
    std::vector<T> gather(const T* per_atom) const               // chain ids, masses: one value per atom of the array
    {
        std::vector<T> r(uniq.size());
        for (size_t k = 0; k < uniq.size(); ++k) r[k] = per_atom[uniq[k]];
        return r;
    }
    void unpack_in_place(uint32_t* atoms, size_t n) const { for (size_t i = 0; i < n; ++i) atoms[i] = uniq[atoms[i]]; }
};

}  // namespace mkamd
