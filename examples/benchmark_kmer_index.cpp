// benchmark_kmer_index.cpp -- the caller of the index path, written against the facade with
// the reference's own spelling. Mirrors the phase sequence of
// test/benchmark/BenchmarkKmerIndex.cpp:401-593 (read query -> sample -> read -> insert ->
// size -> count -> find -> erase) and its compile-time knobs:
//   -DpK=31 -DpDNA=4|5 -DpKmerStore=SINGLE|CANONICAL|BIMOLECULE -DpDistHash=MURMUR|FARM
//   -DpStoreHash=MURMUR|FARM        runtime: -F <fastq> [-Q <query fastq>] [-S <ratio>]
// The query sample is the first n/ratio k-mers (the reference shuffles with
// std::default_random_engine, BenchmarkKmerIndex.cpp:372-392; a fixed prefix keeps runs
// reproducible).
#include <chrono>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "kmerind/kmer_index.hpp"

#define SINGLE 1
#define CANONICAL 2
#define BIMOLECULE 3
#define MURMUR 1
#define FARM 2

#ifndef pK
#define pK 21
#endif
#ifndef pDNA
#define pDNA 4
#endif
#ifndef pKmerStore
#define pKmerStore CANONICAL
#endif
#ifndef pDistHash
#define pDistHash MURMUR
#endif
#ifndef pStoreHash
#define pStoreHash MURMUR
#endif

#if (pDNA == 5)
using Alphabet = bliss::common::DNA5;
#else
using Alphabet = bliss::common::DNA;
#endif
using KmerType = bliss::common::Kmer<pK, Alphabet, bliss::common::WordType>;
using CountType = uint32_t;

#if (pDistHash == FARM)
template <typename KM> using DistHash = bliss::kmer::hash::farm<KM, true>;
#else
template <typename KM> using DistHash = bliss::kmer::hash::murmur<KM, true>;
#endif
#if (pStoreHash == FARM)
template <typename KM> using StoreHash = bliss::kmer::hash::farm<KM, false>;
#else
template <typename KM> using StoreHash = bliss::kmer::hash::murmur<KM, false>;
#endif

#if (pKmerStore == SINGLE)
template <typename Key> using MapParams = ::bliss::index::kmer::SingleStrandHashMapParams<Key, DistHash, StoreHash>;
#elif (pKmerStore == BIMOLECULE)
template <typename Key> using MapParams = ::bliss::index::kmer::BimoleculeHashMapParams<Key, DistHash, StoreHash>;
#else
template <typename Key> using MapParams = ::bliss::index::kmer::CanonicalHashMapParams<Key, DistHash, StoreHash>;
#endif

#if defined(pINDEX_POSQUAL)
// PositionQualityIndex (kmer_index.hpp:405-406): (ShortSequenceKmerId, k-mer quality)
using ValType = std::pair<bliss::common::ShortSequenceKmerId, float>;
using MapType = ::dsc::unordered_multimap<KmerType, ValType, MapParams>;
using IndexType = bliss::index::kmer::PositionQualityIndex<MapType>;
static unsigned long long val_of(const ValType &v) {   // position + the bit pattern of the quality: both are checked by the harness test
  uint32_t b; std::memcpy(&b, &v.second, 4);
  return (unsigned long long)v.first.get_pos() + (unsigned long long)b;
}
#elif defined(pINDEX_POS)
using ValType = bliss::common::ShortSequenceKmerId;
using MapType = ::dsc::unordered_multimap<KmerType, ValType, MapParams>;
using IndexType = bliss::index::kmer::PositionIndex<MapType>;
static unsigned long long val_of(const ValType &v) { return (unsigned long long)v.get_pos(); }
#else
using MapType = ::dsc::counting_unordered_map<KmerType, CountType, MapParams>;
using IndexType = bliss::index::kmer::CountIndex<MapType>;
static unsigned long long val_of(const CountType &v) { return v; }
#endif

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv) {
  std::string filename, queryname;
  int sample_ratio = 100, device = 0;
  for (int i = 1; i + 1 < argc; i += 2) {
    std::string a = argv[i];
    if (a == "-F") filename = argv[i + 1];
    else if (a == "-Q") queryname = argv[i + 1];
    else if (a == "-S") sample_ratio = std::atoi(argv[i + 1]);
    else if (a == "-D") device = std::atoi(argv[i + 1]);
  }
  if (filename.empty()) { std::fprintf(stderr, "usage: %s -F file.fastq [-Q query.fastq] [-S ratio] [-D device]\n", argv[0]); return 2; }
  if (queryname.empty()) queryname = filename;
  if (sample_ratio < 1) sample_ratio = 1;

  try {
    kmerind::comm comm(device);
    IndexType idx(comm);

    // readForQuery_posix (BenchmarkKmerIndex.cpp:311-330): k-mers of the query file
    std::vector<KmerType> query;
    double t = now();
    ::bliss::io::KmerFileHelper::read_file_posix<::bliss::index::kmer::KmerParser<KmerType>, ::bliss::io::FASTQParser,
                                                 ::bliss::io::SequencesIterator>(queryname, query, comm);
    std::printf("[TIME] read_query\t%f\t%zu\n", now() - t, query.size());
    query.resize(query.size() / (size_t)sample_ratio);
    std::printf("[TIME] sample\t0\t%zu\n", query.size());

    {
      std::vector<typename IndexType::KmerParserType::value_type> temp;
      t = now();
      ::bliss::io::KmerFileHelper::read_file_posix<typename IndexType::KmerParserType, ::bliss::io::FASTQParser,
                                                   ::bliss::io::SequencesIterator>(filename, temp, comm);
      std::printf("[TIME] read\t%f\t%zu\n", now() - t, temp.size());
      std::printf("total size is %zu\n", temp.size());
      t = now();
      idx.insert(temp);
      std::printf("[TIME] insert\t%f\t%zu\n", now() - t, idx.local_size());
      std::printf("total size after insert/rehash is %zu\n", idx.size());
    }
    {
      auto lquery = query;
      t = now();
      auto counts = idx.count(lquery);
      size_t present = 0;
      for (auto &c : counts) present += c.second;
      std::printf("[TIME] count\t%f\t%zu\n", now() - t, counts.size());
      std::printf("count results %zu present %zu\n", counts.size(), present);
    }
    {
      auto lquery = query;
      t = now();
      auto found = idx.find(lquery);
      unsigned long long sum = 0;
      for (auto &f : found) sum += val_of(f.second);
      std::printf("[TIME] find\t%f\t%zu\n", now() - t, found.size());
      std::printf("find results %zu sum %llu\n", found.size(), sum);
    }
    {
      auto lquery = query;
      t = now();
      idx.erase(lquery);
      std::printf("[TIME] erase\t%f\t%zu\n", now() - t, idx.local_size());
      std::printf("total size after erase is %zu\n", idx.size());
    }
  } catch (const std::exception &e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  return 0;
}
