// The reference's de Bruijn sample (test/test/test_de_bruijn_graph_construction.cpp) against kmerind/de_bruijn.hpp: the
// same type aliases, build_posix + find for the edge-count and the edge-existence node maps. Where the reference only
// prints sizes, this prints checksums the test harness compares with its checker.
//
//   de_bruijn_graph_construction <file.fastq>
#include <cstdio>
#include <string>
#include <vector>

#include "kmerind/de_bruijn.hpp"

using WordType = uint64_t;
using Alphabet = bliss::common::DNA;
using KmerType = bliss::common::Kmer<21, Alphabet, WordType>;
using EdgeEncoder = bliss::common::DNA16;

template <typename K> using MapParams = ::bliss::index::kmer::BimoleculeHashMapParams<K>;
template <typename EdgeEnc>
using CountNodeMapType = bliss::de_bruijn::de_bruijn_nodes_distributed<KmerType, bliss::de_bruijn::node::edge_counts<EdgeEnc, int32_t>, MapParams>;
template <typename EdgeEnc>
using ExistNodeMapType = bliss::de_bruijn::de_bruijn_nodes_distributed<KmerType, bliss::de_bruijn::node::edge_exists<EdgeEnc>, MapParams>;

template <typename IndexType>
static std::vector<KmerType> readForQuery(const std::string &filename, const kmerind::comm &comm) {
  std::vector<KmerType> query;
  ::bliss::io::KmerFileHelper::template read_file_posix<::bliss::index::kmer::KmerParser<KmerType>, ::bliss::io::FASTQParser,
                                                        ::bliss::io::SequencesIterator>(filename, query, comm);
  return query;
}

static uint64_t word_sum(const KmerType &k) { uint64_t s = 0; for (unsigned w = 0; w < KmerType::nWords; ++w) s += k.getData()[w]; return s; }

template <typename NodeMapType>
static void testDeBruijnGraph(const kmerind::comm &comm, const std::string &filename, const char *tag) {
  NodeMapType idx(comm);
  idx.template build_posix<::bliss::io::FASTQParser, ::bliss::io::SequencesIterator>(filename, comm);
  auto query = readForQuery<NodeMapType>(filename, comm);
  if (query.size() > 50) query.resize(query.size() / 2);          // a part of the input's k-mers, repeats included
  auto results = idx.find(query);
  uint64_t edges = 0, self = 0, keys = 0, nbr = 0;
  for (auto &r : results) {
    keys += word_sum(r.first);
    for (int i = 0; i < 8; ++i) edges += (uint64_t)r.second.get_edge_frequency(i) * (uint64_t)(i + 1);
    std::vector<KmerType> out, in;
    bliss::de_bruijn::node::node_utils<KmerType, typename NodeMapType::ValueType>::get_out_neighbors(r.first, r.second, out);
    bliss::de_bruijn::node::node_utils<KmerType, typename NodeMapType::ValueType>::get_in_neighbors(r.first, r.second, in);
    for (auto &k : out) nbr += word_sum(k) % 1000003ull;
    for (auto &k : in) nbr += word_sum(k) % 1000003ull;
  }
  auto all = idx.to_vector();
  for (auto &r : all) self += (uint64_t)r.second.get_edge_frequency(0) + (uint64_t)r.second.get_edge_frequency(7);
  std::printf("%s nodes %zu size %zu found %zu keysum %llu edgesum %llu nbrsum %llu a_out_t_in %llu\n", tag, idx.local_size(), idx.size(),
              results.size(), (unsigned long long)keys, (unsigned long long)edges, (unsigned long long)nbr, (unsigned long long)self);
  // erase (the node map inherits it from the distributed map): every third query key's node leaves
  std::vector<KmerType> victims;
  for (size_t i = 0; i < query.size(); i += 3) victims.push_back(query[i]);
  const size_t erased = idx.erase(victims);
  uint64_t left = 0;
  for (auto &r : idx.to_vector()) left += word_sum(r.first) % 1000003ull;
  std::printf("%s erased %zu left %zu keysum %llu\n", tag, erased, idx.size(), (unsigned long long)left);
}

int main(int argc, char **argv) {
  if (argc < 2) { std::fprintf(stderr, "usage: %s <file.fastq>\n", argv[0]); return 2; }
  const std::string filename(argv[1]);
  kmerind::comm comm(0);
  try {
    testDeBruijnGraph<bliss::de_bruijn::de_bruijn_engine<CountNodeMapType>>(comm, filename, "count");
    testDeBruijnGraph<bliss::de_bruijn::de_bruijn_engine<ExistNodeMapType>>(comm, filename, "exist");
  } catch (const std::exception &e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  return 0;
}
