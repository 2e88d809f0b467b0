// Compile-and-run check of the map / parameter matrix BenchmarkKmerIndex.cpp instantiates (test/benchmark/BenchmarkKmerIndex.cpp:
// 150-290: pMAP in {SORTED, UNORDERED, DENSEHASH} x pKmerStore x pDistHash x pStoreHash x pDistTrans x pINDEX) through the
// facade's reference names: every Index type builds the same small FASTQ and must hold the same distinct k-mers as the
// first one of its strand model. Needs a GPU to run; compiling it is the drop-in check.
#include <cstdio>
#include <fstream>
#include <string>

#include "kmerind/kmer_index.hpp"

using KmerType = ::bliss::common::Kmer<21, ::bliss::common::DNA, uint64_t>;
using IdType = ::bliss::common::ShortSequenceKmerId;
namespace bik = ::bliss::index::kmer;

template <typename K> using CanonHash = bik::CanonicalHashMapParams<K, bik::DistHashFarm, bik::StoreHashStd>;
template <typename K> using CanonSorted = bik::CanonicalSortedMapParams<K>;
template <typename K> using SingleXor = bik::SingleStrandHashMapParams<K, bik::DistHashMurmur, bik::StoreHashIdentity, ::bliss::kmer::transform::xor_rev_comp>;
template <typename K> using SingleSorted = bik::SingleStrandSortedMapParams<K>;
template <typename K> using BimolHash = bik::BimoleculeHashMapParams<K, bik::DistHashIdentity, bik::StoreHashMurmur>;
template <typename K> using BimolSorted = bik::BimoleculeSortedMapParams<K>;
using Special = ::bliss::kmer::hash::sparsehash::special_keys<KmerType, true>;

static int unsorted_seen = 0;   // a sorted-map flavour whose entries did not come back in key order
template <typename IndexType> static size_t build_size(const std::string &file, bool expect_sorted = false) {
  IndexType idx(::kmerind::comm(0));
  idx.template build_posix<::bliss::io::FASTQParser, ::bliss::io::NSplitSequencesIterator>(file);
  if (expect_sorted) {   // distributed_sorted_map keeps its local container sorted by Less<Key>
    auto v = idx.to_vector();
    for (size_t i = 1; i < v.size(); ++i) if (v[i].first < v[i - 1].first) { unsorted_seen = 1; break; }
  }
  // exists(): one byte per input key; the all-A k-mer is not in the generated file, a stored key is
  std::vector<KmerType> q(2, KmerType(true));
  auto all = idx.to_vector();
  if (!all.empty()) q[1] = all[0].first;
  std::vector<unsigned char> e = idx.exists(q);
  if (e.size() != 2 || (!all.empty() && e[1] != 1)) return 0;
  return idx.local_size();
}

int main(int argc, char **argv) {
  const std::string file = argc > 1 ? argv[1] : "type_matrix.fastq";
  if (argc <= 1) {
    std::ofstream f(file);
    for (int r = 0; r < 40; ++r) {
      std::string seq;
      for (int i = 0; i < 80; ++i) seq += "ACGT"[(r * 7 + i * i + i / 3) & 3];
      if (r % 5 == 0) seq[40] = 'N';
      f << "@r" << r << "\n" << seq << "\n+\n" << std::string(80, 'I') << "\n";
    }
  }
  const size_t c1 = build_size<bik::CountIndex<::dsc::counting_unordered_map<KmerType, uint32_t, CanonHash>>>(file);
  const size_t c2 = build_size<bik::CountIndex<::dsc::counting_densehash_map<KmerType, uint32_t, CanonHash, Special>>>(file);
  const size_t c3 = build_size<bik::CountIndex<::dsc::counting_sorted_map<KmerType, uint32_t, CanonSorted>>>(file, true);
  const size_t s1 = build_size<bik::CountIndex2<::dsc::counting_unordered_map<KmerType, uint32_t, SingleXor>>>(file);
  const size_t s2 = build_size<bik::CountIndex<::dsc::counting_sorted_map<KmerType, uint32_t, SingleSorted>>>(file, true);
  const size_t b1 = build_size<bik::CountIndex<::dsc::counting_unordered_map<KmerType, uint32_t, BimolHash>>>(file);
  const size_t b2 = build_size<bik::CountIndex<::dsc::counting_sorted_map<KmerType, uint32_t, BimolSorted>>>(file, true);
  // saturating_counting_densehash_map<..., uint8_t>: counts stop at 255 (a 2-mer index of this file has counts far above)
  {
    using K2 = ::bliss::common::Kmer<2, ::bliss::common::DNA, uint64_t>;
    using Sp2 = ::bliss::kmer::hash::sparsehash::special_keys<K2, true>;
    bik::CountIndex<::dsc::saturating_counting_densehash_map<K2, uint8_t, CanonHash, Sp2>> sat(::kmerind::comm(0));
    bik::CountIndex<::dsc::counting_densehash_map<K2, uint32_t, CanonHash, Sp2>> plain(::kmerind::comm(0));
    sat.template build_posix<::bliss::io::FASTQParser, ::bliss::io::SequencesIterator>(file);
    plain.template build_posix<::bliss::io::FASTQParser, ::bliss::io::SequencesIterator>(file);
    auto a = sat.to_vector();
    auto b = plain.to_vector();
    bool ok_sat = a.size() == b.size() && !a.empty();
    size_t big = 0;
    for (auto &e : b) {
      for (auto &f : a) if (f.first == e.first) ok_sat = ok_sat && f.second == (e.second > 255u ? 255u : (uint8_t)e.second);
      big += e.second > 255u;
    }
    if (!ok_sat || big == 0) { std::printf("saturating counts MISMATCH\n"); return 1; }
  }
  const size_t p1 = build_size<bik::PositionIndex<::dsc::unordered_multimap<KmerType, IdType, CanonHash>>>(file);
  const size_t p2 = build_size<bik::PositionIndex<::dsc::densehash_multimap<KmerType, IdType, CanonHash, Special>>>(file);
  const size_t p3 = build_size<bik::PositionIndex<::dsc::sorted_multimap<KmerType, IdType, CanonSorted>>>(file, true);
  std::printf("canonical %zu %zu %zu  single %zu %zu  bimolecule %zu %zu  positions %zu %zu %zu\n", c1, c2, c3, s1, s2, b1, b2, p1, p2, p3);
  const bool ok = c1 && c1 == c2 && c2 == c3 && s1 == s2 && b1 == b2 && b1 == c1 && p1 == p2 && p2 == p3 && s1 >= c1 && !unsorted_seen;
  std::printf(ok ? "type matrix ok\n" : "type matrix MISMATCH\n");
  return ok ? 0 : 1;
}
