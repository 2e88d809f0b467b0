// kmerind/de_bruijn.hpp -- the de Bruijn graph engine of the reference (test/test/debruijn/) over the MI355X library.
//
// Same names and meaning as the reference's headers, so test/test/test_de_bruijn_graph_construction.cpp:124-137,195-206
// reads the same:
//
//   using KmerType = bliss::common::Kmer<21, bliss::common::DNA, WordType>;
//   template <typename K> using MapParams = bliss::index::kmer::BimoleculeHashMapParams<K>;
//   template <typename EdgeEnc> using CountNodeMapType = bliss::de_bruijn::de_bruijn_nodes_distributed<
//       KmerType, bliss::de_bruijn::node::edge_counts<EdgeEnc, int32_t>, MapParams>;
//   bliss::de_bruijn::de_bruijn_engine<CountNodeMapType> idx(comm);
//   idx.build_posix<bliss::io::FASTQParser, bliss::io::SequencesIterator>(filename, comm);
//   auto results = idx.find(query);          // std::vector<std::pair<KmerType, edge_counts<DNA16, int32_t>>>
//
// What is different is stated in include/kmerind_hip.h ("de Bruijn graph nodes"): every node is kept under its
// lexicographically smaller strand (the reference keeps the strand that arrived first).
#ifndef KMERIND_DE_BRUIJN_HPP
#define KMERIND_DE_BRUIJN_HPP

#include <array>
#include <ostream>

#include "kmerind/kmer_index.hpp"

namespace bliss {
namespace de_bruijn {

namespace node {
static constexpr unsigned char SENSE = 0;        // de_bruijn_node_trait.hpp:53-54
static constexpr unsigned char ANTI_SENSE = 1;

// de_bruijn_node_trait.hpp:119-131
class input_edge_utils {
 public:
  template <typename Alphabet> static uint8_t reverse_complement_edges(uint8_t const &exts) {
    auto comp = [](uint8_t x) -> uint8_t { return (uint8_t)(((x & 1) << 3) | ((x & 2) << 1) | ((x >> 1) & 2) | ((x >> 3) & 1)); };   // DNA16: bit reversal
    return (uint8_t)((comp(exts & 0xF) << 4) | comp(exts >> 4));
  }
};

// de_bruijn_node_trait.hpp:139-265: [out A C G T; in A C G T; k-mer count]
template <typename ALPHA, typename COUNT = uint32_t>
class edge_counts {
 public:
  using Alphabet = ALPHA;
  using CountType = COUNT;
  static constexpr uint32_t KMI = KMI_DBG_EDGE_COUNTS;
  std::array<COUNT, 9> counts;
  edge_counts() : counts({{0, 0, 0, 0, 0, 0, 0, 0, 0}}) {}
  explicit edge_counts(const uint64_t *words) {   // KMI_DBG_VALUE_WORDS words: uint32_t counts[9] + padding
    const uint32_t *c = reinterpret_cast<const uint32_t *>(words);
    for (int i = 0; i < 9; ++i) counts[i] = (COUNT)c[i];
  }
  COUNT get_edge_frequency(uint8_t idx) const { return idx >= 8 ? 0 : counts[idx]; }
  friend std::ostream &operator<<(std::ostream &ost, const edge_counts &node) {
    ost << " dBGr node: counts self = " << node.counts[8] << " in = [";
    for (int i = 4; i < 8; ++i) ost << node.counts[i] << ",";
    ost << "], out = [";
    for (int i = 0; i < 4; ++i) ost << node.counts[i] << ",";
    return ost << "]";
  }
};

// de_bruijn_node_trait.hpp:269-336: one bit per edge, out A C G T in the low nibble, in A C G T in the high one
template <typename ALPHA>
class edge_exists {
 public:
  using Alphabet = ALPHA;
  using CountType = uint8_t;
  static constexpr uint32_t KMI = KMI_DBG_EDGE_EXISTS;
  uint8_t counts;
  edge_exists() : counts(0) {}
  explicit edge_exists(const uint64_t *words) : counts(0) {
    const uint32_t *c = reinterpret_cast<const uint32_t *>(words);
    for (int i = 0; i < 8; ++i) counts |= (uint8_t)((c[i] ? 1u : 0u) << i);
  }
  uint8_t get_edge_frequency(uint8_t idx) const { return idx >= 8 ? 0 : (counts >> idx) & 0x1; }
  friend std::ostream &operator<<(std::ostream &ost, const edge_exists &node) {
    ost << " dBGr node: in = [";
    for (int i = 4; i < 8; ++i) ost << (int)((node.counts >> i) & 1) << ",";
    ost << "], out = [";
    for (int i = 0; i < 4; ++i) ost << (int)((node.counts >> i) & 1) << ",";
    return ost << "]";
  }
};

// de_bruijn_node_trait.hpp:57-115: the neighbours an edge record names
template <typename Kmer, typename EdgeType>
class node_utils {
 public:
  static void get_out_neighbors(Kmer const &kmer, EdgeType const &edge, std::vector<Kmer> &neighbors) {
    neighbors.clear();
    for (int i = 0; i < 4; ++i)
      if (edge.get_edge_frequency(i) > 0) { neighbors.emplace_back(kmer); neighbors.back().nextFromChar(i); }
  }
  static void get_in_neighbors(Kmer const &kmer, EdgeType const &edge, std::vector<Kmer> &neighbors) {
    neighbors.clear();
    for (int i = 0; i < 4; ++i)
      if (edge.get_edge_frequency(i + 4) > 0) { neighbors.emplace_back(kmer); neighbors.back().nextReverseFromChar(i); }
  }
  static void get_out_neighbors(Kmer const &kmer, EdgeType const &edge, std::vector<std::pair<Kmer, typename EdgeType::CountType>> &neighbors) {
    neighbors.clear();
    for (int i = 0; i < 4; ++i) {
      const typename EdgeType::CountType count = edge.get_edge_frequency(i);
      if (count > 0) { neighbors.emplace_back(kmer, count); neighbors.back().first.nextFromChar(i); }
    }
  }
  static void get_in_neighbors(Kmer const &kmer, EdgeType const &edge, std::vector<std::pair<Kmer, typename EdgeType::CountType>> &neighbors) {
    neighbors.clear();
    for (int i = 0; i < 4; ++i) {
      const typename EdgeType::CountType count = edge.get_edge_frequency(i + 4);
      if (count > 0) { neighbors.emplace_back(kmer, count); neighbors.back().first.nextReverseFromChar(i); }
    }
  }
};
}  // namespace node

// de_bruijn_construct_engine.hpp:90-158: (k-mer, edge byte) tuples; only the DNA16 edge encoder is built
template <typename KmerType, typename EdgeEncoder = ::bliss::common::DNA16>
struct de_bruijn_parser {
  static_assert(std::is_same<EdgeEncoder, ::bliss::common::DNA16>::value, "edges are DNA16 bytes (the ASCII encoder of the reference is commented out there too)");
  using edge_type = uint8_t;
  using value_type = std::pair<KmerType, edge_type>;
  using kmer_type = KmerType;
  static constexpr size_t window_size = KmerType::size;
};

// de_bruijn_nodes_distributed.hpp:57-265 -- the map type names the key, the node type and the hash parameters; the
// storage is the device library's
template <typename Key, typename T, template <typename> class MapParams>
struct de_bruijn_nodes_distributed {
  using key_type = Key;
  using mapped_type = T;
  using value_type = std::pair<Key, T>;
  using params = MapParams<Key>;
  static_assert(params::strand == KMI_STRAND_BIMOLECULE,
                "de bruijn graph does not support transform of input Kmers (de_bruijn_nodes_distributed.hpp:238-239): use BimoleculeHashMapParams");
};

// Index<NodeMap, de_bruijn_parser> as the engine uses it (kmer_index.hpp:100-372 of the reference)
template <typename MapType, typename ParserT>
class NodeIndex {
 public:
  using KmerType = typename MapType::key_type;
  using ValueType = typename MapType::mapped_type;
  using TupleType = std::pair<KmerType, ValueType>;
  using EdgeTuple = typename ParserT::value_type;   // std::pair<Kmer, uint8_t>
  using KmerParserType = ParserT;

  explicit NodeIndex(const ::kmerind::comm &_comm) : comm(_comm) {
    std::memset(&cfg, 0, sizeof(cfg));
    cfg.k = KmerType::size; cfg.alphabet = KmerType::KmerAlphabet::KMI; cfg.strand = KMI_STRAND_BIMOLECULE;
    cfg.dist_hash = MapType::params::dist_hash; cfg.store_hash = MapType::params::store_hash;
    cfg.index_kind = KMI_INDEX_COUNT; cfg.seq_format = KMI_FMT_FASTQ; cfg.seq_filter = KMI_SEQ_ALL; cfg.dist_trans = KMI_DIST_MODEL;
    ::kmerind::check(nullptr, kmi_ctx_create(comm.device, comm.rank(), comm.size(), comm.stream, &ctx));
    ::kmerind::check(ctx, kmi_dbg_create(ctx, &cfg, ValueType::KMI, &g));
    if (comm.size() > 1) {   // collective: every rank constructs
      if (comm.unique_id.size() != KMI_COMM_ID_BYTES)
        throw std::invalid_argument("comm.size() > 1 needs comm.unique_id (kmerind::comm::make_unique_id() on rank 0, handed to every rank)");
      ::kmerind::check(ctx, kmi_comm_create(ctx, comm.unique_id.data(), &rccl));
    } else {
      // KMI_FORCE_DIST=1: a one-rank program goes through the code of size() > 1 (a one-rank RCCL communicator), as Index does
      const char *fd = std::getenv("KMI_FORCE_DIST");
      if (fd && std::atoi(fd) != 0) ::kmerind::check(ctx, kmi_comm_create(ctx, nullptr, &rccl));
    }
  }
  NodeIndex(const NodeIndex &) = delete;
  NodeIndex &operator=(const NodeIndex &) = delete;
  ~NodeIndex() { if (rccl) kmi_comm_destroy(rccl); if (g) kmi_dbg_destroy(g); if (ctx) kmi_ctx_destroy(ctx); }

  // build_posix / build_mmap<FASTQParser, SequencesIterator> (test_de_bruijn_graph_construction.cpp:96)
  template <template <typename> class SeqParser, template <typename, template <typename> class> class SeqIterType>
  void build_posix(const std::string &filename, const ::kmerind::comm & /*comm*/) { build_file<SeqParser, SeqIterType>(filename); }
  template <template <typename> class SeqParser, template <typename, template <typename> class> class SeqIterType>
  void build_posix(const std::string &filename) { build_file<SeqParser, SeqIterType>(filename); }
  template <template <typename> class SeqParser, template <typename, template <typename> class> class SeqIterType>
  void build_mmap(const std::string &filename) { build_file<SeqParser, SeqIterType>(filename); }
  // size() > 1: every rank hands over ITS record-aligned partition of the FASTQ file (collective)
  void build_partition(const uint8_t *bytes, size_t n_bytes) {
    if (!rccl) { ::kmerind::check(ctx, kmi_dbg_build_host(g, bytes, n_bytes)); return; }
    ::kmerind::check(ctx, kmi_dbg_build_dist_host(g, rccl, bytes, n_bytes));
  }

  // insert(std::vector<std::pair<Kmer, uint8_t>>&) (de_bruijn_nodes_distributed.hpp:230-264)
  void insert(std::vector<EdgeTuple> &input) {
    if (comm.size() > 1) throw std::invalid_argument("insert of tuples with size() > 1: use build_partition (the tuples are routed on the device)");
    constexpr unsigned nw = KmerType::nWords;
    std::vector<uint64_t> rec(input.size() * (nw + 1));
    for (size_t i = 0; i < input.size(); ++i) {
      std::memcpy(&rec[i * (nw + 1)], input[i].first.getData(), nw * sizeof(uint64_t));
      rec[i * (nw + 1) + nw] = input[i].second;
    }
    ::kmerind::check(ctx, kmi_dbg_insert_host(g, rec.data(), input.size()));
  }

  // find (test_de_bruijn_graph_construction.cpp:114): one (stored k-mer, node) per distinct query key that is a node
  std::vector<TupleType> find(std::vector<KmerType> &query) const {
    kmi_results r{};
    if (rccl) ::kmerind::check(ctx, kmi_dbg_find_dist_host(g, rccl, ::bliss::index::kmer::detail::words_of(query), query.size(), &r));   // collective
    else ::kmerind::check(ctx, kmi_dbg_find_host(g, ::bliss::index::kmer::detail::words_of(query), query.size(), &r));
    std::vector<TupleType> out;
    out.reserve(r.n);
    for (uint64_t i = 0; i < r.n; ++i)
      out.emplace_back(KmerType(r.keys + i * KmerType::nWords), ValueType(r.values + i * KMI_DBG_VALUE_WORDS));
    kmi_results_free(&r);
    return out;
  }
  std::vector<std::pair<KmerType, size_t>> count(std::vector<KmerType> &query) const {
    kmi_results r{};
    if (rccl) ::kmerind::check(ctx, kmi_dbg_count_dist_host(g, rccl, ::bliss::index::kmer::detail::words_of(query), query.size(), &r));   // collective
    else ::kmerind::check(ctx, kmi_dbg_count_host(g, ::bliss::index::kmer::detail::words_of(query), query.size(), &r));
    std::vector<std::pair<KmerType, size_t>> out(r.n);
    for (uint64_t i = 0; i < r.n; ++i) out[i] = std::make_pair(KmerType(r.keys + i * KmerType::nWords), (size_t)r.values[i]);
    kmi_results_free(&r);
    return out;
  }

  // erase (inherited from the distributed map, distributed_unordered_map.hpp:719-779): the nodes of these k-mers leave the map
  size_t erase(std::vector<KmerType> &query) {
    uint64_t n = 0;
    if (rccl) ::kmerind::check(ctx, kmi_dbg_erase_dist_host(g, rccl, ::bliss::index::kmer::detail::words_of(query), query.size(), &n));   // collective
    else ::kmerind::check(ctx, kmi_dbg_erase_host(g, ::bliss::index::kmer::detail::words_of(query), query.size(), &n));
    return (size_t)n;
  }

  size_t local_size() const { uint64_t n = 0; ::kmerind::check(ctx, kmi_dbg_local_size(g, &n)); return (size_t)n; }
  size_t size() const {
    uint64_t n = 0;
    if (rccl) ::kmerind::check(ctx, kmi_dbg_size_dist(g, rccl, &n)); else ::kmerind::check(ctx, kmi_dbg_local_size(g, &n));
    return (size_t)n;
  }
  void clear() { ::kmerind::check(ctx, kmi_dbg_clear(g)); }
  // this rank's nodes
  std::vector<TupleType> to_vector() const {
    const size_t n = local_size();
    std::vector<uint64_t> keys(n * KmerType::nWords + 1);
    std::vector<uint32_t> c9(n * 9 + 1);
    uint64_t got = 0;
    ::kmerind::check(ctx, kmi_dbg_export_host(g, keys.data(), c9.data(), n, &got));
    std::vector<TupleType> out;
    out.reserve(got);
    for (uint64_t i = 0; i < got; ++i) {
      uint32_t c10[10] = {0};
      std::memcpy(c10, &c9[i * 9], 9 * sizeof(uint32_t));
      out.emplace_back(KmerType(&keys[i * KmerType::nWords]), ValueType(reinterpret_cast<const uint64_t *>(c10)));
    }
    return out;
  }

  kmi_ctx *context() const { return ctx; }

 private:
  template <template <typename> class SeqParser, template <typename, template <typename> class> class SeqIterType>
  void build_file(const std::string &filename) {
    static_assert(SeqIterType<const unsigned char *, SeqParser>::KMI == KMI_SEQ_ALL, "the de Bruijn engine reads every record (SequencesIterator)");
    constexpr uint32_t fmt = SeqParser<const unsigned char *>::KMI;   // FASTQParser (the reference's sample) or FASTAParser
    if (::bliss::index::kmer::detail::format_of(filename) != fmt) throw std::invalid_argument("input filename extension is not supported.");
    ::kmerind::check(ctx, kmi_dbg_set_seq_format(g, fmt));
    if (rccl && fmt != KMI_FMT_FASTQ) throw std::invalid_argument("the de Bruijn engine over ranks reads FASTQ partitions");
    if (rccl) {   // every rank reads its byte range plus look-ahead; the partition is cut at record starts on the device
      for (uint64_t look = 1ull << 20;; look *= 8) {
        ::bliss::index::kmer::detail::FileRange r = ::bliss::index::kmer::detail::read_file_range(filename, comm.rank(), comm.size(), look);
        int need_more = 0;
        ::kmerind::check(ctx, kmi_dbg_build_range_dist_host(g, rccl, r.bytes.data(), r.bytes.size(), r.offset, r.nominal, r.reaches_eof ? 1 : 0, &need_more));
        if (!need_more) return;
      }
    }
    std::vector<uint8_t> bytes = ::bliss::index::kmer::detail::read_whole_file(filename);
    ::kmerind::check(ctx, kmi_dbg_build_host(g, bytes.data(), bytes.size()));
  }

  ::kmerind::comm comm;
  kmi_config cfg;
  kmi_ctx *ctx = nullptr;
  kmi_dbg *g = nullptr;
  kmi_comm *rccl = nullptr;
};

// de_bruijn_construct_engine.hpp:241-242
template <template <typename> class MapType>
using de_bruijn_engine = NodeIndex<MapType<::bliss::common::DNA16>, de_bruijn_parser<typename MapType<::bliss::common::DNA16>::key_type, ::bliss::common::DNA16>>;

}  // namespace de_bruijn
}  // namespace bliss

#endif  // KMERIND_DE_BRUIJN_HPP
