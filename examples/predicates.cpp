// predicates.cpp -- the predicate forms of the index API (kmer_index.hpp:156-194: find_if / count_if / erase_if with
// and without a query vector) and build_posix, on a count index and on a position index, printed so that a test can
// compare the numbers with an independent computation.
//   usage: predicates -F file.fastq [-A file.fasta] [-T threshold] [-D device]
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "kmerind/kmer_index.hpp"

using KmerType = bliss::common::Kmer<21, bliss::common::DNA, bliss::common::WordType>;
template <typename KM> using DistHash = bliss::kmer::hash::murmur<KM, true>;
template <typename KM> using StoreHash = bliss::kmer::hash::murmur<KM, false>;
template <typename Key> using MapParams = ::bliss::index::kmer::CanonicalHashMapParams<Key, DistHash, StoreHash>;
using CountMap = ::dsc::counting_unordered_map<KmerType, uint32_t, MapParams>;
using CountIndexType = bliss::index::kmer::CountIndex<CountMap>;
using IdType = bliss::common::ShortSequenceKmerId;
using PosMap = ::dsc::unordered_multimap<KmerType, IdType, MapParams>;
using PosIndexType = bliss::index::kmer::PositionIndex<PosMap>;

int main(int argc, char **argv) {
  std::string filename, fasta;
  unsigned threshold = 2;
  int device = 0;
  for (int i = 1; i + 1 < argc; i += 2) {
    std::string a = argv[i];
    if (a == "-F") filename = argv[i + 1];
    else if (a == "-A") fasta = argv[i + 1];
    else if (a == "-T") threshold = (unsigned)std::atoi(argv[i + 1]);
    else if (a == "-D") device = std::atoi(argv[i + 1]);
  }
  if (filename.empty()) { std::fprintf(stderr, "usage: %s -F file.fastq [-T threshold] [-D device]\n", argv[0]); return 2; }
  try {
    kmerind::comm comm(device);
    // ---- count index, built straight from the file (Index::build_posix, kmer_index.hpp:239-287)
    CountIndexType idx(comm);
    idx.build_posix<::bliss::io::FASTQParser, ::bliss::io::SequencesIterator>(filename);
    std::printf("count size %zu\n", idx.size());
    auto frequent = [threshold](const std::pair<KmerType, uint32_t> &e) { return e.second >= threshold; };
    auto all_frequent = idx.find_if(frequent);
    unsigned long long sum = 0;
    for (auto &e : all_frequent) sum += e.second;
    std::printf("find_if(pred) %zu sum %llu\n", all_frequent.size(), sum);
    auto cnt_all = idx.count_if(frequent);
    std::printf("count_if(pred) %zu\n", cnt_all.size());

    std::vector<KmerType> query;
    ::bliss::io::KmerFileHelper::read_file_posix<::bliss::index::kmer::KmerParser<KmerType>, ::bliss::io::FASTQParser,
                                                 ::bliss::io::SequencesIterator>(filename, query, comm);
    query.resize(query.size() / 2);
    auto q1 = query;
    auto fq = idx.find_if(q1, frequent);
    sum = 0;
    for (auto &e : fq) sum += e.second;
    std::printf("find_if(query,pred) %zu sum %llu\n", fq.size(), sum);
    auto q2 = query;
    auto cq = idx.count_if(q2, frequent);
    size_t ones = 0;
    for (auto &e : cq) ones += e.second;
    std::printf("count_if(query,pred) %zu present %zu\n", cq.size(), ones);
    auto q3 = query;
    idx.erase_if(q3, frequent);
    std::printf("after erase_if(query,pred) %zu\n", idx.size());
    idx.erase_if(frequent);
    std::printf("after erase_if(pred) %zu\n", idx.size());

    // ---- position index: predicate on the stored id
    PosIndexType pidx(comm);
    pidx.build_posix<::bliss::io::FASTQParser, ::bliss::io::SequencesIterator>(filename);
    std::printf("pos size %zu\n", pidx.size());
    auto odd = [](const std::pair<KmerType, IdType> &e) { return (e.second.get_pos() & 1u) != 0; };
    auto podd = pidx.find_if(odd);
    std::printf("pos find_if(pred) %zu\n", podd.size());
    auto q4 = query;
    pidx.erase_if(q4, odd);
    std::printf("pos after erase_if(query,pred) %zu\n", pidx.size());
    pidx.erase_if(odd);
    std::printf("pos after erase_if(pred) %zu\n", pidx.size());
    // ---- FASTA through the same entry point: build_posix<FASTAParser, ...> (kmer_index.hpp:239-287)
    if (!fasta.empty()) {
      CountIndexType fidx(comm);
      fidx.build_posix<::bliss::io::FASTAParser, ::bliss::io::SequencesIterator>(fasta);
      auto all = fidx.to_vector();
      unsigned long long total = 0;
      for (auto &e : all) total += e.second;
      std::printf("fasta size %zu total %llu\n", fidx.size(), total);
    }
  } catch (const std::exception &e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  return 0;
}
