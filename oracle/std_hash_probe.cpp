// Test infrastructure (see oracle/kmerind_oracle.h): the one third-party function bliss::kmer::hash::cpp_std calls
// (src/index/kmer_hash.hpp:161,178,185: ::std::hash<size_t> op; h = op(h); hp = op(data[i])) taken from the C++ standard
// library of this toolchain itself, so that the oracle's restatement of cpp_std is checked against the real thing rather
// than against the assumption that libstdc++ hashes a size_t to itself.
#include <cstddef>
#include <functional>

extern "C" size_t orc_probe_std_hash_size_t(size_t x) { return ::std::hash<size_t>()(x); }
