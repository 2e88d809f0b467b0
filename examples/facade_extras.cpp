// facade_extras.cpp -- the members of bliss::index::kmer::Index a reference caller reaches beyond the benchmark's
// sequence (src/index/kmer_index.hpp): insert(std::vector<std::pair<Kmer, count>>&) ADDING the values (:200-225 ->
// distributed_unordered_map.hpp:1603-1618), get_map() / cbegin() / cend() (:120-125, 377-384), and the
// PositionQualityIndex alias (:405-406) with its (ShortSequenceKmerId, float) values. Prints numbers the test checks
// in tests/test_gpu_facade.py.
//   usage: facade_extras file.fastq [file.fasta]
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "kmerind/kmer_index.hpp"

using KmerType = bliss::common::Kmer<21, bliss::common::DNA, uint64_t>;
template <typename Key> using MapParams = ::bliss::index::kmer::CanonicalHashMapParams<Key>;
using CountMap = ::dsc::counting_unordered_map<KmerType, uint32_t, MapParams>;
using CountIdx = bliss::index::kmer::CountIndex<CountMap>;
using QVal = std::pair<bliss::common::ShortSequenceKmerId, float>;
using QMap = ::dsc::unordered_multimap<KmerType, QVal, MapParams>;
using QIdx = bliss::index::kmer::PositionQualityIndex<QMap>;

int main(int argc, char **argv) {
  if (argc < 2) { std::fprintf(stderr, "usage: %s file.fastq\n", argv[0]); return 2; }
  const std::string fastq = argv[1];
  try {
    kmerind::comm comm(0);
    // ---- weighted pairs: the count parser's tuples carry 1; insert them, then the same keys again with weight 3
    std::vector<std::pair<KmerType, uint32_t>> tuples;
    ::bliss::io::KmerFileHelper::read_file_posix<typename CountIdx::KmerParserType, ::bliss::io::FASTQParser, ::bliss::io::SequencesIterator>(fastq, tuples, comm);
    CountIdx cidx(comm);
    {
      auto t = tuples;
      cidx.insert(t);                                     // every occurrence with weight 1
      for (auto &e : t) e.second = 3;
      cidx.insert(t);                                     // + 3 per occurrence
    }
    unsigned long long sum = 0, n = 0;
    for (auto it = cidx.cbegin(); it != cidx.cend(); ++it) { sum += it->second; ++n; }   // the map walked through its iterators
    std::printf("weighted entries %llu sum %llu occurrences %zu\n", n, sum, tuples.size());
    {
      // update (distributed_densehash_map.hpp:1975-2030): pairs of stored keys add their value and report 1, unknown keys are skipped
      std::vector<std::pair<KmerType, uint32_t>> upd;
      for (size_t i = 0; i < tuples.size(); i += 11) upd.push_back(std::make_pair(tuples[i].first, 5u));
      upd.push_back(std::make_pair(KmerType(), 7u));   // AAAA...A: not in these reads
      const size_t hit = cidx.update(upd, false, [](uint32_t &stored, uint32_t const &v) { stored += v; return 1; });
      // filter form: every entry with a count of at least 20 is halved
      const size_t halved = cidx.update([](const std::pair<KmerType, uint32_t> &e) { return e.second >= 20; }, [](uint32_t &stored) { stored /= 2; return 1; });
      unsigned long long s2 = 0;
      for (auto &e : cidx.to_vector()) s2 += e.second;
      std::printf("update hit %zu halved %zu sum %llu entries %zu\n", hit, halved, s2, cidx.local_size());
      // the same call with device-side updaters (kmerind::updater): max with 9 on every 7th occurrence, then assign -- the
      // pairs of one key are applied in input order, so its LAST pair stays (value = input position mod 1000)
      std::vector<std::pair<KmerType, uint32_t>> up2, up3;
      for (size_t i = 0; i < tuples.size(); i += 7) up2.push_back(std::make_pair(tuples[i].first, 9u));
      for (size_t i = 0; i < tuples.size(); i += 3) up3.push_back(std::make_pair(tuples[i].first, (uint32_t)(up3.size() % 1000)));
      up3.push_back(std::make_pair(KmerType(), 1u));
      const size_t h2 = cidx.update(up2, false, kmerind::updater::max());
      unsigned long long s3 = 0;
      for (auto &e : cidx.to_vector()) s3 += e.second;
      const size_t h3 = cidx.update(up3, false, kmerind::updater::assign());
      unsigned long long s4 = 0;
      for (auto &e : cidx.to_vector()) s4 += e.second;
      std::printf("device max hit %zu sum %llu assign hit %zu sum %llu\n", h2, s3, h3, s4);
    }
    {
      // Index::build_posix (kmer_index.hpp:239-372): the file entry point itself; with KMI_FORCE_DIST=1 this is the path of
      // comm.size() > 1 -- the rank's byte range plus look-ahead, record-aligned on the device, the collective build
      CountIdx fidx(comm);
      fidx.build_posix<::bliss::io::FASTQParser, ::bliss::io::SequencesIterator>(fastq);
      unsigned long long fs = 0;
      for (auto &e : fidx.to_vector()) fs += e.second;
      std::vector<KmerType> q;
      for (size_t i = 0; i < tuples.size(); i += 4) q.push_back(tuples[i].first);
      q.push_back(KmerType());
      unsigned long long ex = 0;
      for (unsigned char b : fidx.exists(q)) ex += b;
      std::printf("build_posix entries %zu sum %llu exists %llu of %zu\n", fidx.local_size(), fs, ex, q.size());
    }
    if (argc > 2) {
      // the same entry point on a FASTA file (with KMI_FORCE_DIST=1: every rank reads the file whole and keeps its block of an
      // equal split, the block bookkeeping computed on the device), and read_file_posix on it
      const std::string fasta = argv[2];
      CountIdx aidx(comm);
      aidx.build_posix<::bliss::io::FASTAParser, ::bliss::io::SequencesIterator>(fasta);
      unsigned long long as = 0;
      for (auto &e : aidx.to_vector()) as += e.second;
      std::vector<std::pair<KmerType, uint32_t>> at;
      ::bliss::io::KmerFileHelper::read_file_posix<typename CountIdx::KmerParserType, ::bliss::io::FASTAParser, ::bliss::io::SequencesIterator>(fasta, at, comm);
      std::printf("fasta build_posix entries %zu sum %llu tuples %zu\n", aidx.local_size(), as, at.size());
    }
    auto &view = cidx.get_map();
    std::printf("get_map local_size %zu size %zu\n", view.local_size(), view.size());
    // ---- PositionQualityIndex: (k-mer, (id, quality)) tuples through read_file + insert, then find
    std::vector<std::pair<KmerType, QVal>> qt;
    ::bliss::io::KmerFileHelper::read_file_posix<typename QIdx::KmerParserType, ::bliss::io::FASTQParser, ::bliss::io::SequencesIterator>(fastq, qt, comm);
    QIdx qidx(comm);
    { auto t = qt; qidx.insert(t); }
    std::vector<KmerType> q;
    for (size_t i = 0; i < qt.size(); i += 5) q.push_back(qt[i].first);
    auto found = qidx.find(q);
    unsigned long long pos = 0, qbits = 0;
    for (auto &f : found) { pos += f.second.first.get_pos(); uint32_t b; std::memcpy(&b, &f.second.second, 4); qbits += b; }
    std::printf("posqual tuples %zu entries %zu found %zu pos %llu qbits %llu\n", qt.size(), qidx.local_size(), found.size(), pos, qbits);
    unsigned long long all_q = 0;
    for (auto &e : qidx.to_vector()) { uint32_t b; std::memcpy(&b, &e.second.second, 4); all_q += b; }
    std::printf("posqual all qbits %llu\n", all_q);
  } catch (const std::exception &e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  return 0;
}
