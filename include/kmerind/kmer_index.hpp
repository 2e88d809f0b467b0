// kmerind/kmer_index.hpp -- header-only C++ facade over libkmerind_hip.so that keeps the
// reference's spelling for the k-mer index path, so code written against
//   bliss::common::Kmer<K, Alphabet, Word>                    (src/common/kmer.hpp:116-177)
//   bliss::index::kmer::Index<MapType, KmerParser>            (src/index/kmer_index.hpp:98-394)
//   bliss::index::kmer::CountIndex / CountIndex2 / KmerIndex  (src/index/kmer_index.hpp:399-411)
//   ::dsc::counting_unordered_map / counting_densehash_map    (src/containers/distributed_*_map.hpp)
//   bliss::index::kmer::{SingleStrand,Canonical,Bimolecule}HashMapParams (kmer_index.hpp:436-481)
//   bliss::io::KmerFileHelper::read_file_posix / _mmap        (src/io/kmer_file_helper.hpp:588-633)
// compiles against this header and runs on an MI355X. Only what sits on that path is
// mirrored; everything is host C++ that forwards to the C ABI in ../kmerind_hip.h.
//
// Differences a maintainer must know (also in INTEGRATION.md):
//   * mxx::comm is replaced by kmerind::comm {device, rank, size}; with size > 1 the exchange runs over RCCL inside the
//     library (comm.unique_id: rank 0's ncclUniqueId bytes, handed to every rank by the application), or through the
//     caller's own all-to-all (kmerind::comm::exchange) when that functor is set.
//   * MapType template arguments are tag types: they only select the kmi_config.
//   * insert()/count()/find()/erase() leave their argument vector unchanged (the reference
//     leaves it in an unspecified state).
//   * errors: KMI_ERR_INVALID -> std::invalid_argument, KMI_ERR_PARSE -> std::logic_error,
//     anything else -> std::runtime_error, like the reference's exception types.
#ifndef KMERIND_KMER_INDEX_HPP
#define KMERIND_KMER_INDEX_HPP

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <limits>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include <map>
#include <algorithm>
#include <vector>

#include "../kmerind_hip.h"

// ---------------------------------------------------------------------------
// communicator stand-in
// ---------------------------------------------------------------------------
namespace kmerind {

struct comm {
  int device = 0;
  int rank_ = 0;
  int size_ = 1;
  void *stream = nullptr;
  // all-to-all of k-mer words between ranks, required only when size() > 1:
  // (send words grouped by destination, send counts in k-mers [size], n_words) -> received words
  std::function<std::vector<uint64_t>(const std::vector<uint64_t> &, const std::vector<uint64_t> &, uint32_t)> exchange;
  std::function<uint64_t(uint64_t)> allreduce_sum;
  // size() > 1 without the two functors above: the exchange runs inside the library over RCCL (kmi_comm_*, grouped
  // ncclSend / ncclRecv over device buffers). All ranks need the same 128-byte id: rank 0 calls make_unique_id() and the
  // application hands the bytes to the others (MPI_Bcast in the reference's world) before the Index is constructed.
  std::vector<char> unique_id;
  // ... or the application's own messenger in place of RCCL (kmi_comm_create_transport): two collective callbacks over host buffers,
  // e.g. MPI_Alltoallv / MPI_Allreduce on the communicator the reference's program already has. Every
  // member then runs the library's own multi-rank code, staged through pinned memory around the callbacks.
  kmi_transport transport{nullptr, nullptr, nullptr};
  static std::vector<char> make_unique_id() {
    std::vector<char> id(KMI_COMM_ID_BYTES);
    if (kmi_comm_unique_id(id.data()) != KMI_OK) throw std::runtime_error("kmerind_hip: RCCL is not available (kmi_comm_unique_id)");
    return id;
  }

  comm() = default;
  explicit comm(int dev, int r = 0, int s = 1) : device(dev), rank_(r), size_(s) {}
  int rank() const { return rank_; }
  int size() const { return size_; }
  void barrier() const {}
};

// updaters of Index::update that run on the device (kmi_index_update_pairs_*): op(stored, v) = stored + v, max(stored, v),
// min(stored, v), v -- each returning 1, so update() returns the number of input pairs whose key is stored
namespace updater {
struct add { static constexpr uint32_t KMI = KMI_UPDATE_ADD; };
struct max { static constexpr uint32_t KMI = KMI_UPDATE_MAX; };
struct min { static constexpr uint32_t KMI = KMI_UPDATE_MIN; };
struct assign { static constexpr uint32_t KMI = KMI_UPDATE_ASSIGN; };
}  // namespace updater

inline void check(kmi_ctx *ctx, kmi_status st) {
  if (st == KMI_OK) return;
  std::string msg = std::string("kmerind_hip: ") + (ctx ? kmi_last_error(ctx) : "error");
  if (st == KMI_ERR_INVALID) throw std::invalid_argument(msg);
  if (st == KMI_ERR_PARSE) throw std::logic_error(msg);
  throw std::runtime_error(msg + " (status " + std::to_string((int)st) + ")");
}

}  // namespace kmerind

// ---------------------------------------------------------------------------
// bliss::common : alphabets and the packed k-mer value type
// ---------------------------------------------------------------------------
namespace bliss {
namespace common {

using WordType = uint64_t;  // base_types.hpp:36

struct DNA { static constexpr unsigned SIZE = 4; static constexpr unsigned BITS = 2; static constexpr uint32_t KMI = KMI_ALPHA_DNA; };
struct DNA6 { static constexpr unsigned SIZE = 8; static constexpr unsigned BITS = 3; static constexpr uint32_t KMI = KMI_ALPHA_DNA5; };
using DNA5 = DNA6;  // alphabets.hpp:746-747
struct RNA { static constexpr unsigned SIZE = 4; static constexpr unsigned BITS = 2; static constexpr uint32_t KMI = KMI_ALPHA_RNA; };
struct RNA6 { static constexpr unsigned SIZE = 8; static constexpr unsigned BITS = 3; static constexpr uint32_t KMI = KMI_ALPHA_RNA5; };
using RNA5 = RNA6;  // alphabets.hpp:748-750
struct DNA16 { static constexpr unsigned SIZE = 16; static constexpr unsigned BITS = 4; static constexpr uint32_t KMI = KMI_ALPHA_DNA16; };

template <typename A> struct AlphabetTraits {
  static constexpr unsigned getSize() { return A::SIZE; }
  static constexpr unsigned getBitsPerChar() { return A::BITS; }
};

// Same object layout as the reference: WORD_TYPE data[nWords], data[0] least significant,
// newest base in the low bits, pad bits zero, no alignment attribute (kmer.hpp:177).
template <unsigned int KMER_SIZE, typename ALPHABET, typename WORD_TYPE = WordType>
class Kmer {
  static_assert(std::is_same<WORD_TYPE, uint64_t>::value, "the MI355X path stores k-mers in 64-bit words");

 public:
  static constexpr unsigned int size = KMER_SIZE;
  static constexpr unsigned int bitsPerChar = ALPHABET::BITS;
  static constexpr unsigned int nBits = size * bitsPerChar;
  static constexpr unsigned int nWords = (nBits + 63) / 64;
  static constexpr unsigned int nBytes = (nBits + 7) / 8;
  using KmerWordType = WORD_TYPE;
  using KmerAlphabet = ALPHABET;

  explicit Kmer(bool clear = true) { if (clear) std::memset(data, 0, sizeof(data)); }
  explicit Kmer(const WORD_TYPE *words) { std::memcpy(data, words, sizeof(data)); sanitize(); }

  const WORD_TYPE *getData() const { return data; }
  WORD_TYPE *getDataRef() { return data; }
  const WORD_TYPE *getConstData() const { return data; }

  // Kmer::nextFromChar (kmer.hpp:731-741): c is an alphabet code, not ASCII
  void nextFromChar(unsigned char c) {
    for (int w = (int)nWords - 1; w > 0; --w) data[w] = (data[w] << bitsPerChar) | (data[w - 1] >> (64 - bitsPerChar));
    data[0] = (data[0] << bitsPerChar) | (static_cast<WORD_TYPE>(c) & ((WORD_TYPE(1) << bitsPerChar) - 1));
    sanitize();
  }
  // Kmer::nextReverseFromChar (kmer.hpp:758-768): the window moves one base to the left, c becomes the oldest base
  void nextReverseFromChar(unsigned char c) {
    for (unsigned w = 0; w + 1 < nWords; ++w) data[w] = (data[w] >> bitsPerChar) | (data[w + 1] << (64 - bitsPerChar));
    data[nWords - 1] >>= bitsPerChar;
    constexpr unsigned top = (size - 1) * bitsPerChar;
    data[top / 64] |= (static_cast<WORD_TYPE>(c) & ((WORD_TYPE(1) << bitsPerChar) - 1)) << (top % 64);
    if ((top % 64) + bitsPerChar > 64) data[top / 64 + 1] |= (static_cast<WORD_TYPE>(c) & ((WORD_TYPE(1) << bitsPerChar) - 1)) >> (64 - top % 64);
    sanitize();
  }
  void sanitize() {
    constexpr unsigned inv_pad = 64 - (nWords * 64 - nBits);
    if (inv_pad < 64) data[nWords - 1] &= ((WORD_TYPE(1) << inv_pad) - 1);
  }

  bool operator==(const Kmer &o) const { return std::memcmp(data, o.data, sizeof(data)) == 0; }
  bool operator!=(const Kmer &o) const { return !(*this == o); }
  bool operator<(const Kmer &o) const {  // kmer.hpp:820-823: compare from the most significant word
    for (int w = (int)nWords - 1; w >= 0; --w)
      if (data[w] != o.data[w]) return data[w] < o.data[w];
    return false;
  }
  bool operator>(const Kmer &o) const { return o < *this; }
  bool operator<=(const Kmer &o) const { return !(o < *this); }
  bool operator>=(const Kmer &o) const { return !(*this < o); }

 private:
  WORD_TYPE data[nWords];
};

// sequence.hpp:127-209 / :231-296 -- same single-word layout and accessors
struct ShortSequenceKmerId {
  size_t id = 0;
  ShortSequenceKmerId() = default;
  explicit ShortSequenceKmerId(size_t raw) : id(raw) {}
  size_t get_id() const { return (id >> 16) & 0x000000FFFFFFFFFFull; }
  size_t get_pos() const { return get_id() + (id & 0xFFFF); }
  uint8_t get_file_id() const { return (uint8_t)(id >> 56); }
  bool operator==(const ShortSequenceKmerId &o) const { return id == o.id; }
  bool operator<(const ShortSequenceKmerId &o) const { return id < o.id; }
};
struct LongSequenceKmerId {
  size_t id = 0;
  LongSequenceKmerId() = default;
  explicit LongSequenceKmerId(size_t raw) : id(raw) {}
  size_t get_id() const { return (id >> 40) & 0xFFFF; }
  size_t get_pos() const { return id & 0x000000FFFFFFFFFFull; }
  uint8_t get_file_id() const { return (uint8_t)(id >> 56); }
  bool operator==(const LongSequenceKmerId &o) const { return id == o.id; }
  bool operator<(const LongSequenceKmerId &o) const { return id < o.id; }
};

}  // namespace common

// transform / hash tags (kmer_transform.hpp:90-145, kmer_hash.hpp:242-311)
namespace transform { template <typename K> struct identity {}; }
namespace kmer {
// lex_greater (kmer_transform.hpp:128-145) picks the larger strand. No *MapParams of kmer_index.hpp uses it; it is accepted as
// a DistTrans / StoreTrans tag and treated as lex_less (as sets of stored keys the two differ by which strand represents a pair).
namespace transform { template <typename K> struct lex_less {}; template <typename K> struct xor_rev_comp {}; template <typename K> struct lex_greater {}; }
namespace hash {
template <typename K, bool Prefix = false> struct murmur { static constexpr uint32_t KMI = KMI_HASH_MURMUR; };
template <typename K, bool Prefix = false> struct farm { static constexpr uint32_t KMI = KMI_HASH_FARM; };
template <typename K, bool Prefix = false> struct identity { static constexpr uint32_t KMI = KMI_HASH_IDENTITY; };
template <typename K, bool Prefix = false> struct cpp_std { static constexpr uint32_t KMI = KMI_HASH_STD; };
// the empty / deleted key pair of the densehash maps (kmer_hash.hpp sparsehash::special_keys): storage detail, a tag here
namespace sparsehash { template <typename K, bool Canonical = false> struct special_keys {}; }
}  // namespace hash
}  // namespace kmer

// sequence / tuple parser tags (fastq_loader.hpp, fasta_loader.hpp, kmer_parser.hpp)
namespace io {
template <typename Iter> struct FASTQParser { static constexpr uint32_t KMI = KMI_FMT_FASTQ; };
template <typename Iter> struct FASTAParser { static constexpr uint32_t KMI = KMI_FMT_FASTA; };
// the SeqIterType argument of read_file_* / build_* (sequence_iterator.hpp:96-300, filtered_sequence_iterator.hpp:154-165,429-440)
template <typename Iter, template <typename> class Parser> struct SequencesIterator { static constexpr uint32_t KMI = KMI_SEQ_ALL; };
template <typename Iter, template <typename> class Parser> struct NFilterSequencesIterator { static constexpr uint32_t KMI = KMI_SEQ_N_FILTER; };
template <typename Iter, template <typename> class Parser> struct NSplitSequencesIterator { static constexpr uint32_t KMI = KMI_SEQ_N_SPLIT; };
}  // namespace io
}  // namespace bliss

// ---------------------------------------------------------------------------
// ::dsc map tags (distributed_map_base.hpp:87-143 DistributedMapParams)
// ---------------------------------------------------------------------------
namespace dsc {

template <typename Key, template <typename> class InputTrans, template <typename> class DistTrans,
          template <typename> class DistHash, template <typename> class DistEqual, template <typename> class StoreTrans,
          template <typename> class StoreHash, template <typename> class StoreEqual>
struct HashMapParams {
  static constexpr bool input_is_lex_less = std::is_same<InputTrans<Key>, ::bliss::kmer::transform::lex_less<Key>>::value;
  static constexpr bool store_is_lex_less = std::is_same<StoreTrans<Key>, ::bliss::kmer::transform::lex_less<Key>>::value;
  static constexpr uint32_t strand = input_is_lex_less ? KMI_STRAND_CANONICAL : (store_is_lex_less ? KMI_STRAND_BIMOLECULE : KMI_STRAND_SINGLE);
  static constexpr uint32_t dist_hash = DistHash<Key>::KMI;
  static constexpr uint32_t store_hash = StoreHash<Key>::KMI;
  // DistTrans of the single-strand model ("could be iden, xor, lex_less", kmer_index.hpp:436-450); the other models fix it
  static constexpr uint32_t dist_trans = strand != KMI_STRAND_SINGLE ? KMI_DIST_MODEL :
      (std::is_same<DistTrans<Key>, ::bliss::kmer::transform::lex_less<Key>>::value ? KMI_DIST_LEX :
       (std::is_same<DistTrans<Key>, ::bliss::kmer::transform::xor_rev_comp<Key>>::value ? KMI_DIST_XOR : KMI_DIST_MODEL));
};

template <typename Key, typename T, template <typename> class MapParams>
struct counting_unordered_map {
  using key_type = Key; using mapped_type = T; using params = MapParams<Key>;
  static constexpr uint32_t index_kind = KMI_INDEX_COUNT;
  static constexpr bool saturating = false;   // std::plus<T>: a count type narrower than the device's 32 bits wraps
  static constexpr bool sorted = false;       // the sorted maps hand their local entries out in key order
};
// saturating_counting_densehash_map (distributed_densehash_map.hpp:2903-2953, sat_plus<T>): counts stop at the largest T.
// The device counts in 32 bits; what a caller reads is min(count, max(T)), which is what a chain of sat_plus(+1) gives.
template <typename Key, typename T, template <typename> class MapParams, typename SpecialKeys = void>
struct saturating_counting_densehash_map : counting_unordered_map<Key, T, MapParams> {
  static constexpr bool saturating = true;
};
template <typename Key, typename T, template <typename> class MapParams, typename SpecialKeys = void>
struct counting_densehash_map : counting_unordered_map<Key, T, MapParams> {};
// multimaps (distributed_unordered_map.hpp:1466-1515): T is the position id type
template <typename Key, typename T, template <typename> class MapParams>
struct unordered_multimap {
  using key_type = Key; using mapped_type = T; using params = MapParams<Key>;
  // T = ShortSequenceKmerId / LongSequenceKmerId (one word) or std::pair<ShortSequenceKmerId, float> (kmer_index.hpp:405-406:
  // two words on the device, the float's bits in the low half of the second)
  static constexpr uint32_t index_kind = sizeof(T) == sizeof(uint64_t) ? KMI_INDEX_POSITION : KMI_INDEX_POSQUAL;
  static constexpr bool saturating = false;
  static constexpr bool sorted = false;
  static_assert(sizeof(T) == sizeof(uint64_t) || sizeof(T) == 2 * sizeof(uint64_t), "values are one (id) or two ((id, quality)) 64-bit words");
};
template <typename Key, typename T, template <typename> class MapParams, typename SpecialKeys = void>
struct densehash_multimap : unordered_multimap<Key, T, MapParams> {};

// sorted flavours (distributed_sorted_map.hpp; pMAP == SORTED in BenchmarkKmerIndex.cpp:196-218): the reference keeps a
// sorted vector per rank and splits ranks by sample sort; the map a caller observes -- count / find / erase / size /
// to_vector as a set -- is the unordered one's, so the tags select the same device index. (With more than one rank the
// owner of a key follows the hash rule here, not a sort split.)
template <typename Key, template <typename> class InputTrans, template <typename> class StoreTrans,
          template <typename> class Less, template <typename> class Equal>
struct SortedMapParams {
  static constexpr bool input_is_lex_less = std::is_same<InputTrans<Key>, ::bliss::kmer::transform::lex_less<Key>>::value;
  static constexpr bool store_is_lex_less = std::is_same<StoreTrans<Key>, ::bliss::kmer::transform::lex_less<Key>>::value;
  static constexpr uint32_t strand = input_is_lex_less ? KMI_STRAND_CANONICAL : (store_is_lex_less ? KMI_STRAND_BIMOLECULE : KMI_STRAND_SINGLE);
  static constexpr uint32_t dist_hash = KMI_HASH_MURMUR, store_hash = KMI_HASH_MURMUR, dist_trans = KMI_DIST_MODEL;
};
template <typename Key, typename T, template <typename> class MapParams>
struct counting_sorted_map : counting_unordered_map<Key, T, MapParams> { static constexpr bool sorted = true; };
// (distributed_sorted_map.hpp keeps the local container a vector sorted by Less<Key>: what to_vector / cbegin..cend of these
// tags walk is in ascending key order -- Kmer::operator< for the std::less the *SortedMapParams aliases default to --, equal
// keys of the multimap in the order the device holds them. The range PARTITION of keys over ranks by sampled splitters is
// not reproduced: which rank holds a key is internal here as it is there.)
template <typename Key, typename T, template <typename> class MapParams>
struct sorted_multimap : unordered_multimap<Key, T, MapParams> { static constexpr bool sorted = true; };

}  // namespace dsc

// ---------------------------------------------------------------------------
// bliss::index::kmer
// ---------------------------------------------------------------------------
namespace bliss {
namespace index {
namespace kmer {

template <typename K> using DistHashMurmur = ::bliss::kmer::hash::murmur<K, true>;
template <typename K> using DistHashFarm = ::bliss::kmer::hash::farm<K, true>;
template <typename K> using StoreHashMurmur = ::bliss::kmer::hash::murmur<K, false>;
template <typename K> using StoreHashFarm = ::bliss::kmer::hash::farm<K, false>;
template <typename K> using DistHashStd = ::bliss::kmer::hash::cpp_std<K, true>;          // kmer_index.hpp:416-419
template <typename K> using DistHashIdentity = ::bliss::kmer::hash::identity<K, true>;
template <typename K> using StoreHashStd = ::bliss::kmer::hash::cpp_std<K, false>;        // :426-429
template <typename K> using StoreHashIdentity = ::bliss::kmer::hash::identity<K, false>;

template <typename Key, template <typename> class DistHash = DistHashMurmur, template <typename> class StoreHash = StoreHashMurmur,
          template <typename> class DistTrans = ::bliss::transform::identity>
using SingleStrandHashMapParams = ::dsc::HashMapParams<Key, ::bliss::transform::identity, DistTrans, DistHash, ::std::equal_to,
                                                      ::bliss::transform::identity, StoreHash, ::std::equal_to>;
template <typename Key, template <typename> class DistHash = DistHashMurmur, template <typename> class StoreHash = StoreHashMurmur>
using CanonicalHashMapParams = ::dsc::HashMapParams<Key, ::bliss::kmer::transform::lex_less, ::bliss::transform::identity, DistHash,
                                                   ::std::equal_to, ::bliss::transform::identity, StoreHash, ::std::equal_to>;
template <typename Key, template <typename> class DistHash = DistHashMurmur, template <typename> class StoreHash = StoreHashMurmur>
using BimoleculeHashMapParams = ::dsc::HashMapParams<Key, ::bliss::transform::identity, ::bliss::kmer::transform::lex_less, DistHash,
                                                    ::std::equal_to, ::bliss::kmer::transform::lex_less, StoreHash, ::std::equal_to>;

// kmer_index.hpp:530-562
template <typename Key, template <typename> class Less = ::std::less>
using SingleStrandSortedMapParams = ::dsc::SortedMapParams<Key, ::bliss::transform::identity, ::bliss::transform::identity, Less, ::std::equal_to>;
template <typename Key, template <typename> class Less = ::std::less>
using CanonicalSortedMapParams = ::dsc::SortedMapParams<Key, ::bliss::kmer::transform::lex_less, ::bliss::transform::identity, Less, ::std::equal_to>;
template <typename Key, template <typename> class Less = ::std::less>
using BimoleculeSortedMapParams = ::dsc::SortedMapParams<Key, ::bliss::transform::identity, ::bliss::kmer::transform::lex_less, Less, ::std::equal_to>;

// tuple parsers (kmer_parser.hpp:85-294, 909-1083): value_type is what read_file_* produces
template <typename KmerType> struct KmerParser { using value_type = KmerType; using kmer_type = KmerType; static constexpr size_t window_size = KmerType::size; };
template <typename TupleType> struct KmerCountTupleParser {
  using value_type = TupleType; using kmer_type = typename std::tuple_element<0, TupleType>::type;
  static constexpr size_t window_size = kmer_type::size;
};

template <typename TupleType> struct KmerPositionQualityTupleParser {   // kmer_parser.hpp:577-900: (k-mer, (ShortSequenceKmerId, quality))
  using value_type = TupleType; using kmer_type = typename std::tuple_element<0, TupleType>::type;
  static constexpr size_t window_size = kmer_type::size;
};
template <typename TupleType> struct KmerPositionTupleParser {   // kmer_parser.hpp:303-569
  using value_type = TupleType; using kmer_type = typename std::tuple_element<0, TupleType>::type;
  static constexpr size_t window_size = kmer_type::size;
};

namespace detail {
template <typename MapType> kmi_config make_config(uint32_t fmt) {
  using Key = typename MapType::key_type;
  kmi_config c;
  c.k = Key::size; c.alphabet = Key::KmerAlphabet::KMI; c.strand = MapType::params::strand;
  c.dist_hash = MapType::params::dist_hash; c.store_hash = MapType::params::store_hash;
  c.index_kind = MapType::index_kind; c.seq_format = fmt; c.farm_ndebug = 0; c.seq_filter = KMI_SEQ_ALL; c.dist_trans = MapType::params::dist_trans;
  return c;
}
inline std::vector<uint8_t> read_whole_file(const std::string &filename) {
  FILE *f = std::fopen(filename.c_str(), "rb");
  if (!f) throw std::invalid_argument("cannot open " + filename);
  std::fseek(f, 0, SEEK_END); long n = std::ftell(f); std::fseek(f, 0, SEEK_SET);
  std::vector<uint8_t> buf((size_t)(n > 0 ? n : 0));
  if (n > 0 && std::fread(buf.data(), 1, (size_t)n, f) != (size_t)n) { std::fclose(f); throw std::runtime_error("short read on " + filename); }
  std::fclose(f);
  return buf;
}
// what ONE rank of p reads of a file (partitioned_file, file.hpp:1216-1430): its nominal byte range [n r / p, n (r + 1) / p) plus
// `lookahead` bytes (where its last record ends and the next rank's first begins); *reaches_eof: the buffer ends with the file
struct FileRange { std::vector<uint8_t> bytes; uint64_t offset = 0, nominal = 0; bool reaches_eof = false; int prev_byte = -1 /* the file byte before the range, -1 at the file start */; };
inline FileRange read_file_range(const std::string &filename, int rank, int p, uint64_t lookahead) {
  FILE *f = std::fopen(filename.c_str(), "rb");
  if (!f) throw std::invalid_argument("cannot open " + filename);
  std::fseek(f, 0, SEEK_END);
  const uint64_t n = (uint64_t)std::ftell(f);
  const uint64_t lo = n / (uint64_t)p * (uint64_t)rank + (n % (uint64_t)p) * (uint64_t)rank / (uint64_t)p;
  const uint64_t hi = (rank + 1 == p) ? n : n / (uint64_t)p * (uint64_t)(rank + 1) + (n % (uint64_t)p) * (uint64_t)(rank + 1) / (uint64_t)p;
  uint64_t end = hi + lookahead;
  if (end > n || end < hi) end = n;
  FileRange r;
  r.offset = lo; r.nominal = hi - lo; r.reaches_eof = end == n;
  r.bytes.resize((size_t)(end - lo));
  if (lo > 0) { std::fseek(f, (long)(lo - 1), SEEK_SET); r.prev_byte = std::fgetc(f); }
  std::fseek(f, (long)lo, SEEK_SET);
  if (end > lo && std::fread(r.bytes.data(), 1, (size_t)(end - lo), f) != (size_t)(end - lo)) { std::fclose(f); throw std::runtime_error("short read on " + filename); }
  std::fclose(f);
  return r;
}
inline uint32_t format_of(const std::string &filename) {  // kmer_index.hpp:243-254
  auto ends = [&](const char *e) { size_t l = std::strlen(e); return filename.size() >= l && filename.compare(filename.size() - l, l, e) == 0; };
  if (ends(".fastq") || ends(".fq")) return KMI_FMT_FASTQ;
  if (ends(".fasta") || ends(".fa")) return KMI_FMT_FASTA;
  throw std::invalid_argument("input filename extension is not supported.");
}
template <typename V> typename std::enable_if<std::is_arithmetic<V>::value, V>::type value_of(uint64_t w) { return (V)w; }
// the stored count as the map's mapped_type: wrapped (std::plus) or stopped at the type's maximum (sat_plus)
template <typename MapType, typename V> typename std::enable_if<std::is_arithmetic<V>::value, V>::type count_of(uint64_t w) {
  if (MapType::saturating && w > (uint64_t)std::numeric_limits<V>::max()) return std::numeric_limits<V>::max();
  return (V)w;
}
template <typename MapType, typename V> typename std::enable_if<!std::is_arithmetic<V>::value, V>::type count_of(uint64_t w) { return value_of<V>(w); }
template <typename V> typename std::enable_if<!std::is_arithmetic<V>::value, V>::type value_of(uint64_t w) { return V((size_t)w); }
// multimap values <-> device words: an id is one word; (id, quality) is two, the float's bits in the low half of the second
template <typename V> struct value_words { static constexpr unsigned N = 1;
  static void to(const V &v, uint64_t *w) { uint64_t t = 0; std::memcpy(&t, &v, sizeof(V) < sizeof(uint64_t) ? sizeof(V) : sizeof(uint64_t)); w[0] = t; }
  static V from(const uint64_t *w) { return value_of<V>(w[0]); } };
template <typename Id> struct value_words<std::pair<Id, float>> { static constexpr unsigned N = 2;
  static void to(const std::pair<Id, float> &v, uint64_t *w) { std::memcpy(w, &v.first, sizeof(uint64_t)); uint32_t b; std::memcpy(&b, &v.second, 4); w[1] = b; }
  static std::pair<Id, float> from(const uint64_t *w) { uint32_t b = (uint32_t)w[1]; float q; std::memcpy(&q, &b, 4); return std::make_pair(Id((size_t)w[0]), q); } };
// what find() / to_vector() hand back for one entry: the count as mapped_type, or the multimap value from its words
template <typename MapType, typename V> typename std::enable_if<MapType::index_kind == KMI_INDEX_COUNT, V>::type stored_value(const uint64_t *w) { return count_of<MapType, V>(w[0]); }
template <typename MapType, typename V> typename std::enable_if<MapType::index_kind != KMI_INDEX_COUNT, V>::type stored_value(const uint64_t *w) { return value_words<V>::from(w); }
template <typename MapType, typename V> constexpr unsigned stored_words() { return MapType::index_kind == KMI_INDEX_COUNT ? 1u : value_words<V>::N; }
template <typename Kmer> const uint64_t *words_of(const std::vector<Kmer> &v) { return reinterpret_cast<const uint64_t *>(v.data()); }
template <typename Kmer, typename T> std::vector<uint64_t> words_of_pairs(const std::vector<std::pair<Kmer, T>> &v) {
  std::vector<uint64_t> w(v.size() * Kmer::nWords);
  for (size_t i = 0; i < v.size(); ++i) std::memcpy(&w[i * Kmer::nWords], v[i].first.getData(), sizeof(uint64_t) * Kmer::nWords);
  return w;
}
}  // namespace detail

template <typename MapType, typename KmerParserT>
class Index {
 public:
  using KmerType = typename MapType::key_type;
  using ValueType = typename MapType::mapped_type;
  using TupleType = std::pair<KmerType, ValueType>;
  using Alphabet = typename KmerType::KmerAlphabet;
  using KmerParserType = KmerParserT;
  static_assert(sizeof(KmerType) == KmerType::nWords * sizeof(uint64_t), "Kmer must be a plain word array");

  explicit Index(const ::kmerind::comm &_comm) : comm(_comm) {
    cfg = detail::make_config<MapType>(KMI_FMT_FASTQ);
    ::kmerind::check(nullptr, kmi_ctx_create(comm.device, comm.rank(), comm.size(), comm.stream, &ctx));
    ::kmerind::check(ctx, kmi_index_create(ctx, &cfg, &idx));
    // sat_plus of a 32-bit count runs on the device (narrower count types saturate when they are read, count_of)
    if (MapType::saturating && MapType::index_kind == KMI_INDEX_COUNT) ::kmerind::check(ctx, kmi_index_set_saturating(idx, 1));
    if (comm.size() > 1 && comm.transport.all_to_all_v) {   // the library's collectives over the application's messenger
      ::kmerind::check(ctx, kmi_comm_create_transport(ctx, &comm.transport, &rccl));
    } else if (comm.size() > 1 && !comm.exchange) {   // the exchange runs inside the library over RCCL (collective: every rank constructs)
      if (comm.unique_id.size() != KMI_COMM_ID_BYTES)
        throw std::invalid_argument("comm.size() > 1 needs comm.unique_id (kmerind::comm::make_unique_id() on rank 0, handed to every rank), comm.transport or comm.exchange");
      ::kmerind::check(ctx, kmi_comm_create(ctx, comm.unique_id.data(), &rccl));
    } else if (comm.size() == 1) {
      // KMI_FORCE_DIST=1 (rehearsals on one GPU): a one-rank communicator, and every member takes the path it takes over ranks
      const char *fd = std::getenv("KMI_FORCE_DIST");
      if (fd && std::atoi(fd) != 0) ::kmerind::check(ctx, kmi_comm_create(ctx, nullptr, &rccl));
    }
  }
  Index(const Index &) = delete;
  Index &operator=(const Index &) = delete;
  virtual ~Index() { if (rccl) kmi_comm_destroy(rccl); if (idx) kmi_index_destroy(idx); if (ctx) kmi_ctx_destroy(ctx); }

  // Index::insert (kmer_index.hpp:200-225): vector<Kmer>, or the map's own tuples. For the counting maps a tuple's value is
  // ADDED to the key's count (reduction_unordered_map::local_insert, distributed_unordered_map.hpp:1603-1618); for the
  // multimaps every (k-mer, value) is kept.
  void insert(std::vector<KmerType> &temp) { insert_words(detail::words_of(temp), temp.size()); }
  void insert(std::vector<TupleType> &temp) { insert_tuples(temp, std::integral_constant<bool, MapType::index_kind == KMI_INDEX_COUNT>()); }

  // Index::count (:142-145): one (key, 0|1) per distinct transformed query key
  std::vector<std::pair<KmerType, size_t>> count(std::vector<KmerType> &query) const {
    kmi_results r{};
    if (rccl) ::kmerind::check(ctx, kmi_index_count_dist_host(idx, rccl, detail::words_of(query), query.size(), &r));
    else {
      std::vector<uint64_t> q = route_queries(query);
      ::kmerind::check(ctx, kmi_index_count_host(idx, q.data(), q.size() / KmerType::nWords, &r));
    }
    // (the device hands count results back at the stride of the map's values: one word, two for (id, quality))
    constexpr unsigned ow = detail::stored_words<MapType, ValueType>();
    std::vector<std::pair<KmerType, size_t>> out(r.n);
    for (uint64_t i = 0; i < r.n; ++i) out[i] = std::make_pair(KmerType(r.keys + i * KmerType::nWords), (size_t)r.values[i * ow]);
    kmi_results_free(&r);
    return out;
  }
  // exists() of the densehash maps (distributed_densehash_map.hpp:1465-1560): one byte per INPUT key, in input order
  // (1 = stored). The device answers per distinct transformed key; the bytes are filled on the host from that answer.
  std::vector<unsigned char> exists(std::vector<KmerType> &query) const {
    std::vector<unsigned char> out(query.size(), 0);
    const size_t nw = KmerType::nWords;
    // count() answers per distinct transformed key, over ranks too (the query keys travel to their owners and the answers come
    // back to the rank that asked); with size() > 1 every rank calls, possibly with an empty vector
    std::vector<std::pair<KmerType, size_t>> cnt = count(query);
    if (query.empty()) return out;
    std::sort(cnt.begin(), cnt.end(), [](const std::pair<KmerType, size_t> &a, const std::pair<KmerType, size_t> &b) { return a.first < b.first; });
    std::vector<uint64_t> t(query.size() * nw);
    const uint64_t *w = detail::words_of(query);
    if (cfg.strand == KMI_STRAND_SINGLE) std::memcpy(t.data(), w, t.size() * sizeof(uint64_t));
    else ::kmerind::check(ctx, kmi_canonical_host(ctx, &cfg, w, query.size(), t.data()));   // the key the map stores
    for (size_t i = 0; i < query.size(); ++i) {
      const KmerType key(&t[i * nw]);
      auto it = std::lower_bound(cnt.begin(), cnt.end(), key, [](const std::pair<KmerType, size_t> &a, const KmerType &b) { return a.first < b; });
      out[i] = (it != cnt.end() && !(key < it->first) && it->second != 0) ? 1 : 0;
    }
    return out;
  }
  // Index::find (:132-135): (key, stored value) of present query keys
  std::vector<TupleType> find(std::vector<KmerType> &query) const {
    kmi_results r{};
    if (rccl) ::kmerind::check(ctx, kmi_index_find_dist_host(idx, rccl, detail::words_of(query), query.size(), &r));
    else {
      std::vector<uint64_t> q = route_queries(query);
      ::kmerind::check(ctx, kmi_index_find_host(idx, q.data(), q.size() / KmerType::nWords, &r));
    }
    constexpr unsigned ow = detail::stored_words<MapType, ValueType>();
    std::vector<TupleType> out(r.n);
    for (uint64_t i = 0; i < r.n; ++i) out[i] = std::make_pair(KmerType(r.keys + i * KmerType::nWords), detail::stored_value<MapType, ValueType>(r.values + i * ow));
    kmi_results_free(&r);
    return out;
  }
  // Index::erase (:147-149)
  void erase(std::vector<KmerType> &query) {
    uint64_t n = 0;
    if (rccl) { ::kmerind::check(ctx, kmi_index_erase_dist_host(idx, rccl, detail::words_of(query), query.size(), &n)); return; }
    std::vector<uint64_t> q = route_queries(query);
    ::kmerind::check(ctx, kmi_index_erase_host(idx, q.data(), q.size() / KmerType::nWords, &n));
  }

  // Predicate forms (kmer_index.hpp:156-194 -> map.find/count/erase(query, false, pred) and (pred)). The predicate is a
  // host functor on a stored (key, value) entry, so it is evaluated on the host over the entries the device returns;
  // the result sets are the reference's. The forms without a query work on this rank's entries, as in the reference.
  template <typename Predicate> std::vector<TupleType> find_if(std::vector<KmerType> &query, Predicate const &pred) const {
    return filtered(find(query), pred);
  }
  template <typename Predicate> std::vector<TupleType> find_if(Predicate const &pred) const { return filtered(to_vector(), pred); }
  template <typename Predicate>
  std::vector<std::pair<KmerType, size_t>> count_if(std::vector<KmerType> &query, Predicate const &pred) const {
    std::vector<std::pair<KmerType, size_t>> out = count(query);           // one entry per distinct transformed key
    std::map<KmerType, size_t> hits;
    for (const TupleType &e : filtered(find(query), pred)) ++hits[e.first];
    for (auto &kv : out) { auto it = hits.find(kv.first); kv.second = (it == hits.end()) ? 0 : it->second; }
    return out;
  }
  template <typename Predicate> std::vector<std::pair<KmerType, size_t>> count_if(Predicate const &pred) const {
    std::map<KmerType, size_t> hits;
    for (const TupleType &e : filtered(to_vector(), pred)) ++hits[e.first];
    return std::vector<std::pair<KmerType, size_t>>(hits.begin(), hits.end());
  }
  template <typename Predicate> void erase_if(std::vector<KmerType> &query, Predicate const &pred) {
    std::vector<TupleType> found = find(query);
    erase_entries(found, pred);
  }
  template <typename Predicate> void erase_if(Predicate const &pred) {
    std::vector<TupleType> all = to_vector();
    erase_entries(all, pred);
  }

  // update() of the densehash maps (distributed_densehash_map.hpp:1975-2030 -> densehash_map.hpp:663-736): for every input
  // pair whose (transformed) key is stored, `count += op(stored_value, input_value)`, op changing the stored value in
  // place; keys that are not stored are skipped. The filter form visits every stored entry fop accepts with
  // `count += op(stored_value)`. Updater and Filter are host functors, so -- like the predicate forms -- they run on
  // the host over the entries the device returns; the changed values go back as (key, value) pairs. Counting maps.
  template <typename Updater> size_t update(std::vector<TupleType> &input, bool /*sorted_input*/, Updater const &op) {
    static_assert(MapType::index_kind == KMI_INDEX_COUNT, "update() is a member of the counting / reduction maps");
    constexpr unsigned nw = KmerType::nWords;
    if (comm.size() > 1 || rccl) {
      // over ranks (distributed_densehash_map.hpp:1975-2030: distribute, then the local update): the pairs travel to the owners of
      // their keys (kmi_index_route_pairs_dist_host; collective, an empty input still enters it), and every owner applies the
      // functor on its host to ITS entries (pairs of one key arrive grouped by source rank; inside a source their order is the
      // routing scatter's, so a functor that is not commutative sees an unspecified order there) and writes the changed values back. The return value counts what the functor returned HERE, as the reference's local update does.
      need_rccl("update");
      std::vector<uint64_t> rec(input.size() * (nw + 1) + 1);
      for (size_t i = 0; i < input.size(); ++i) {
        std::memcpy(&rec[i * (nw + 1)], input[i].first.getData(), sizeof(uint64_t) * nw);
        rec[i * (nw + 1) + nw] = weight_word(input[i])[0];
      }
      kmi_results mine{};
      ::kmerind::check(ctx, kmi_index_route_pairs_dist_host(idx, rccl, rec.data(), input.size(), &mine));
      size_t count = 0;
      if (mine.n && local_size()) {
        // this rank's entries for the keys that arrived (a local lookup: the keys are stored keys of this rank already)
        kmi_results r{};
        ::kmerind::check(ctx, kmi_index_find_host(idx, mine.keys, mine.n, &r));
        constexpr unsigned ow = detail::stored_words<MapType, ValueType>();
        std::map<KmerType, ValueType> cur;
        for (uint64_t i = 0; i < r.n; ++i) cur[KmerType(r.keys + i * nw)] = detail::stored_value<MapType, ValueType>(r.values + i * ow);
        kmi_results_free(&r);
        std::map<KmerType, bool> touched;
        for (uint64_t i = 0; i < mine.n; ++i) {
          auto it = cur.find(KmerType(mine.keys + i * nw));
          if (it == cur.end()) continue;
          count += (size_t)op(it->second, detail::count_of<MapType, ValueType>(mine.values[i]));
          touched[it->first] = true;
        }
        write_back(cur, touched);
      }
      kmi_results_free(&mine);
      return count;
    }
    if (input.empty() || local_size() == 0) return 0;
    std::vector<KmerType> keys(input.size());
    for (size_t i = 0; i < input.size(); ++i) keys[i] = input[i].first;
    std::vector<TupleType> found = find(keys);
    std::map<KmerType, ValueType> cur;
    for (const TupleType &e : found) cur[e.first] = e.second;
    std::vector<uint64_t> t(input.size() * nw);   // the keys as the map stores them
    if (cfg.strand == KMI_STRAND_SINGLE) std::memcpy(t.data(), detail::words_of(keys), t.size() * sizeof(uint64_t));
    else ::kmerind::check(ctx, kmi_canonical_host(ctx, &cfg, detail::words_of(keys), keys.size(), t.data()));
    size_t count = 0;
    std::map<KmerType, bool> touched;
    for (size_t i = 0; i < input.size(); ++i) {
      auto it = cur.find(KmerType(&t[i * nw]));
      if (it == cur.end()) continue;
      count += (size_t)op(it->second, input[i].second);
      touched[it->first] = true;
    }
    write_back(cur, touched);
    return count;
  }
  // ... with one of the arithmetic updaters of kmerind::updater the whole call runs on the device: the pairs are partitioned
  // like queries, every bucket's entries sit in an LDS table and the counts are updated in place (kmi_update.h)
  size_t update(std::vector<TupleType> &input, bool, ::kmerind::updater::add const &) { return update_on_device(input, KMI_UPDATE_ADD); }
  size_t update(std::vector<TupleType> &input, bool, ::kmerind::updater::max const &) { return update_on_device(input, KMI_UPDATE_MAX); }
  size_t update(std::vector<TupleType> &input, bool, ::kmerind::updater::min const &) { return update_on_device(input, KMI_UPDATE_MIN); }
  size_t update(std::vector<TupleType> &input, bool, ::kmerind::updater::assign const &) { return update_on_device(input, KMI_UPDATE_ASSIGN); }
  template <typename Filter, typename Updater> size_t update(Filter const &fop, Updater const &op) {
    static_assert(MapType::index_kind == KMI_INDEX_COUNT, "update() is a member of the counting / reduction maps");
    std::map<KmerType, ValueType> cur;
    std::map<KmerType, bool> touched;
    size_t count = 0;
    for (TupleType &e : to_vector()) {
      if (!fop(e)) continue;
      count += (size_t)op(e.second);
      cur[e.first] = e.second; touched[e.first] = true;
    }
    write_back(cur, touched);
    return count;
  }

  size_t local_size() const { uint64_t n = 0; ::kmerind::check(ctx, kmi_index_local_size(idx, &n)); return (size_t)n; }
  size_t size() const {
    if (rccl) { uint64_t n = 0; ::kmerind::check(ctx, kmi_index_size_dist(idx, rccl, &n)); return (size_t)n; }
    size_t n = local_size();
    return (comm.size() > 1 && comm.allreduce_sum) ? (size_t)comm.allreduce_sum(n) : n;
  }
  // get_map() / cbegin() / cend() (kmer_index.hpp:120-125, 377-384): the reference hands out its local container and
  // iterators into it. The entries live in HBM here, so the "map" a caller can walk is a host snapshot of this rank's
  // entries (to_vector), refreshed on every get_map() / cbegin() call; cend() belongs to the snapshot cbegin() took.
  struct MapView {
    const Index *owner; std::vector<TupleType> entries;
    using const_iterator = typename std::vector<TupleType>::const_iterator;
    const_iterator cbegin() const { return entries.cbegin(); }
    const_iterator cend() const { return entries.cend(); }
    const_iterator begin() const { return entries.cbegin(); }
    const_iterator end() const { return entries.cend(); }
    size_t local_size() const { return entries.size(); }
    size_t size() const { return owner->size(); }
    std::vector<TupleType> to_vector() const { return entries; }
  };
  MapView &get_map() { view.owner = this; view.entries = to_vector(); return view; }
  typename MapView::const_iterator cbegin() { return get_map().cbegin(); }
  typename MapView::const_iterator cend() const { return view.cend(); }

  // MapType::to_vector (distributed_map_base.hpp:202-217)
  std::vector<TupleType> to_vector() const {
    uint64_t n = local_size(), got = 0;
    if (MapType::index_kind != KMI_INDEX_COUNT) {
      constexpr unsigned vw = detail::stored_words<MapType, ValueType>();
      std::vector<uint64_t> keys(n * KmerType::nWords + 1), vals(n * vw + 1);
      ::kmerind::check(ctx, kmi_index_export_tuples_host(idx, keys.data(), vals.data(), n, &got));
      std::vector<TupleType> out(got);
      for (uint64_t i = 0; i < got; ++i) out[i] = std::make_pair(KmerType(&keys[i * KmerType::nWords]), detail::stored_value<MapType, ValueType>(&vals[i * vw]));
      if (MapType::sorted) std::stable_sort(out.begin(), out.end(), [](const TupleType &a, const TupleType &b) { return a.first < b.first; });
      return out;
    }
    std::vector<uint64_t> keys(n * KmerType::nWords + 1);
    std::vector<uint32_t> cnt(n + 1);
    ::kmerind::check(ctx, kmi_index_export_host(idx, keys.data(), cnt.data(), n, &got));
    std::vector<TupleType> out(got);
    for (uint64_t i = 0; i < got; ++i) { const uint64_t c = cnt[i]; out[i] = std::make_pair(KmerType(&keys[i * KmerType::nWords]), detail::stored_value<MapType, ValueType>(&c)); }
    if (MapType::sorted) std::sort(out.begin(), out.end(), [](const TupleType &a, const TupleType &b) { return a.first < b.first; });
    return out;
  }

  // Index::build_posix / build_mmap (:239-372): read_file + insert; the extension check is the reference's
  template <template <typename> class SeqParser, template <typename, template <typename> class> class SeqIterType>
  void build_posix(const std::string &filename, void * /*MPI_Comm*/ = nullptr) {
    build_file<SeqParser>(filename, SeqIterType<const unsigned char *, SeqParser>::KMI);
  }
  template <template <typename> class SeqParser, template <typename, template <typename> class> class SeqIterType>
  void build_mmap(const std::string &filename, void * = nullptr) { build_file<SeqParser>(filename, SeqIterType<const unsigned char *, SeqParser>::KMI); }
  // build_mpiio (:239-262): the reference reads through MPI-IO; which call brings the bytes into host memory is not observable
  template <template <typename> class SeqParser, template <typename, template <typename> class> class SeqIterType>
  void build_mpiio(const std::string &filename, void * = nullptr) { build_file<SeqParser>(filename, SeqIterType<const unsigned char *, SeqParser>::KMI); }
  // size() > 1: every rank passes ITS record-aligned partition of the file (bytes + the partition's offset in the file; the
  // partition negotiation of file.hpp:1216-1430 is the caller's, e.g. kmerind_amd.fileio or an MPI-IO split): parse, route,
  // exchange over RCCL, insert. Collective.
  void build_partition(const uint8_t *bytes, size_t n_bytes, uint64_t file_offset, uint32_t fmt = KMI_FMT_FASTQ, uint32_t seq_filter = KMI_SEQ_ALL) {
    ::kmerind::check(ctx, kmi_index_set_seq_format(idx, fmt));
    ::kmerind::check(ctx, kmi_index_set_seq_filter(idx, seq_filter));
    if (comm.size() == 1 && !rccl) { ::kmerind::check(ctx, kmi_index_build_host(idx, bytes, n_bytes, file_offset)); return; }
    need_rccl("build_partition");
    ::kmerind::check(ctx, kmi_index_build_dist_host(idx, rccl, bytes, n_bytes, file_offset));
  }

  kmi_ctx *context() const { return ctx; }
  const kmi_config &config() const { return cfg; }

 protected:
  size_t update_on_device(std::vector<TupleType> &input, uint32_t op) {
    static_assert(MapType::index_kind == KMI_INDEX_COUNT, "update() is a member of the counting / reduction maps");
    if (comm.size() > 1) need_rccl("update");
    if (input.empty() && comm.size() == 1) return 0;
    constexpr unsigned nw = KmerType::nWords;
    std::vector<uint64_t> rec(input.size() * (nw + 1) + 1);
    for (size_t i = 0; i < input.size(); ++i) {
      std::memcpy(&rec[i * (nw + 1)], input[i].first.getData(), sizeof(uint64_t) * nw);
      rec[i * (nw + 1) + nw] = (uint64_t)input[i].second & 0xffffffffull;
    }
    uint64_t n = 0;
    // over ranks the pairs travel to the owners of their keys first (collective); the return value counts the pairs applied HERE
    if (rccl) ::kmerind::check(ctx, kmi_index_update_pairs_dist_host(idx, rccl, rec.data(), input.size(), op, &n));
    else ::kmerind::check(ctx, kmi_index_update_pairs_host(idx, rec.data(), input.size(), op, &n));
    return (size_t)n;
  }
  template <typename Predicate> static std::vector<TupleType> filtered(std::vector<TupleType> v, Predicate const &pred) {
    v.erase(std::remove_if(v.begin(), v.end(), [&](const TupleType &e) { return !pred(e); }), v.end());
    return v;
  }
  // erase the entries of `entries` (all entries of their keys, as find returns them) that satisfy pred. The device
  // erases by key; for a multimap the entries of those keys that do not satisfy pred are put back.
  template <typename Predicate> void erase_entries(const std::vector<TupleType> &entries, Predicate const &pred) {
    std::vector<KmerType> keys;
    std::vector<TupleType> keep;
    std::map<KmerType, bool> hit;
    for (const TupleType &e : entries) if (pred(e)) hit[e.first] = true;
    if (hit.empty()) return;
    for (auto &kv : hit) keys.push_back(kv.first);
    if (MapType::index_kind != KMI_INDEX_COUNT)
      for (const TupleType &e : entries) if (hit.count(e.first) && !pred(e)) keep.push_back(e);
    uint64_t n = 0;
    // entries hold stored (already transformed) keys of this rank: no routing, no second transform needed
    ::kmerind::check(ctx, kmi_index_erase_host(idx, detail::words_of(keys), keys.size(), &n));
    if (!keep.empty()) insert(keep);
  }
  template <template <typename> class SeqParser> void build_file(const std::string &filename, uint32_t seq_filter = KMI_SEQ_ALL) {
    uint32_t fmt = detail::format_of(filename);
    if (fmt != SeqParser<const unsigned char *>::KMI) throw std::invalid_argument("Specified File Parser template parameter does not support files with this extension.");
    ::kmerind::check(ctx, kmi_index_set_seq_format(idx, fmt));
    ::kmerind::check(ctx, kmi_index_set_seq_filter(idx, seq_filter));
    if (comm.size() > 1 || rccl) {
      // every rank reads ITS byte range of the file plus look-ahead, finds where its partition begins and ends with the
      // four-line rule on the device (both neighbours decide at the same file position) and enters the collective build
      // (partitioned_file + FASTQParser::find_first_record, file.hpp:1216-1430, fastq_loader.hpp:269-364)
      need_rccl("build_posix / build_mmap / build_mpiio");
      if (fmt == KMI_FMT_FASTA) {
        // FASTA: the rank's block of the equal split plus look-ahead; what the block cannot know from its own bytes -- which record
        // it starts in, in which state -- comes from the other ranks' block summaries, one small gather inside the library
        // (fasta_loader.hpp:202-470 gets it from collectives over the blocks' first and last lines)
        for (uint64_t look = 1ull << 16;; look *= 16) {
          detail::FileRange r = detail::read_file_range(filename, comm.rank(), comm.size(), look);
          int need_more = 0;
          ::kmerind::check(ctx, kmi_index_build_fasta_range_dist_host(idx, rccl, r.bytes.data(), r.bytes.size(), r.offset, r.nominal, r.reaches_eof ? 1 : 0,
                                                                       r.prev_byte, &need_more));
          if (!need_more) return;
        }
      }
      if (fmt != KMI_FMT_FASTQ) throw std::invalid_argument("build_* with size() > 1 reads FASTQ or FASTA files");
      for (uint64_t look = 1ull << 20;; look *= 8) {
        detail::FileRange r = detail::read_file_range(filename, comm.rank(), comm.size(), look);
        int need_more = 0;
        ::kmerind::check(ctx, kmi_index_build_range_dist_host(idx, rccl, r.bytes.data(), r.bytes.size(), r.offset, r.nominal, r.reaches_eof ? 1 : 0, &need_more));
        if (!need_more) return;
      }
    }
    std::vector<uint8_t> bytes = detail::read_whole_file(filename);
    ::kmerind::check(ctx, kmi_index_build_host(idx, bytes.data(), bytes.size(), 0));
  }
  void insert_words(const uint64_t *words, size_t n) {
    if (rccl) { ::kmerind::check(ctx, kmi_index_insert_dist_host(idx, rccl, words, n)); return; }
    if (comm.size() == 1) { ::kmerind::check(ctx, kmi_index_insert_host(idx, words, n)); return; }
    std::vector<uint64_t> mine = route_words(words, n);
    ::kmerind::check(ctx, kmi_index_insert_host(idx, mine.data(), mine.size() / KmerType::nWords));
  }
  // imxx::distribute for size() > 1: KeyToRank on the device, all-to-all by the caller's exchange
  std::vector<uint64_t> route_words(const uint64_t *words, size_t n) const {
    if (!comm.exchange) throw std::invalid_argument("comm.size() > 1 needs comm.exchange (all-to-all of k-mer words)");
    const uint32_t p = (uint32_t)comm.size(), nw = KmerType::nWords;
    std::vector<uint64_t> canon(n * nw);
    std::vector<uint32_t> ranks(n);
    if (n) {
      if (cfg.strand == KMI_STRAND_SINGLE) std::memcpy(canon.data(), words, sizeof(uint64_t) * n * nw);
      else ::kmerind::check(ctx, kmi_canonical_host(ctx, &cfg, words, n, canon.data()));
      ::kmerind::check(ctx, kmi_key_to_rank_host(ctx, &cfg, canon.data(), n, p, ranks.data()));
    }
    std::vector<uint64_t> counts(p, 0), offs(p, 0), send(n * nw);
    for (size_t i = 0; i < n; ++i) ++counts[ranks[i]];
    for (uint32_t r = 1; r < p; ++r) offs[r] = offs[r - 1] + counts[r - 1];
    for (size_t i = 0; i < n; ++i) std::memcpy(&send[(offs[ranks[i]]++) * nw], &canon[i * nw], sizeof(uint64_t) * nw);  // stable
    return comm.exchange(send, counts, nw);
  }
  std::vector<uint64_t> route_queries(const std::vector<KmerType> &query) const {
    const uint64_t *w = detail::words_of(query);
    if (comm.size() == 1) return std::vector<uint64_t>(w, w + query.size() * KmerType::nWords);
    return route_words(w, query.size());
  }

  void insert_tuples(std::vector<TupleType> &temp, std::true_type /* counting map */) {
    auto w = detail::words_of_pairs(temp);
    constexpr unsigned nw = KmerType::nWords;
    if (comm.size() > 1 && !rccl) {
      // (through comm.exchange only keys travel: weights other than 1 need the library's own exchange)
      for (const TupleType &t : temp)
        if (weight_word(t)[0] != 1ull) throw std::invalid_argument("weighted (k-mer, count) insert with size() > 1 needs the RCCL communicator (comm.unique_id)");
      insert_words(w.data(), temp.size());
      return;
    }
    std::vector<uint64_t> rec(temp.size() * (nw + 1) + 1);
    for (size_t i = 0; i < temp.size(); ++i) {
      std::memcpy(&rec[i * (nw + 1)], &w[i * nw], sizeof(uint64_t) * nw);
      rec[i * (nw + 1) + nw] = weight_word(temp[i])[0] & 0xffffffffull;
    }
    if (rccl) ::kmerind::check(ctx, kmi_index_insert_pairs_dist_host(idx, rccl, rec.data(), temp.size()));   // the pairs go to their keys' owners
    else ::kmerind::check(ctx, kmi_index_insert_pairs_host(idx, rec.data(), temp.size()));
  }
  void insert_tuples(std::vector<TupleType> &temp, std::false_type /* multimap: (k-mer, position id[, quality]) */) {
    auto w = detail::words_of_pairs(temp);
    constexpr unsigned vw = detail::value_words<ValueType>::N;
    std::vector<uint64_t> vals(temp.size() * vw + 1);
    for (size_t i = 0; i < temp.size(); ++i) detail::value_words<ValueType>::to(temp[i].second, &vals[i * vw]);
    if (comm.size() == 1 && !rccl) { ::kmerind::check(ctx, kmi_index_insert_tuples_host(idx, w.data(), vals.data(), temp.size())); return; }
    need_rccl("multimap insert");
    ::kmerind::check(ctx, kmi_index_insert_tuples_dist_host(idx, rccl, w.data(), vals.data(), temp.size()));
  }
  // the touched entries leave the map and come back with their new values (stored keys: no second transform needed)
  void write_back(const std::map<KmerType, ValueType> &cur, const std::map<KmerType, bool> &touched) {
    if (touched.empty()) return;
    constexpr unsigned nw = KmerType::nWords;
    std::vector<KmerType> keys;
    std::vector<uint64_t> rec;
    for (auto &kv : touched) {
      keys.push_back(kv.first);
      const uint64_t *w = kv.first.getData();
      rec.insert(rec.end(), w, w + nw);
      rec.push_back((uint64_t)cur.at(kv.first) & 0xffffffffull);
    }
    uint64_t n = 0;
    ::kmerind::check(ctx, kmi_index_erase_host(idx, detail::words_of(keys), keys.size(), &n));
    ::kmerind::check(ctx, kmi_index_insert_pairs_host(idx, rec.data(), keys.size()));
  }
  void need_rccl(const char *what) const {
    if (!rccl) throw std::invalid_argument(std::string(what) + " with size() > 1 needs the RCCL communicator (comm.unique_id)");
  }
  static std::vector<uint64_t> weight_word(const TupleType &t) {
    std::vector<uint64_t> w(detail::value_words<ValueType>::N + 1, 0);
    std::memcpy(w.data(), &t.second, sizeof(t.second) < sizeof(uint64_t) ? sizeof(t.second) : sizeof(uint64_t));
    return w;
  }

  ::kmerind::comm comm;
  kmi_config cfg;
  kmi_ctx *ctx = nullptr;
  kmi_index *idx = nullptr;
  kmi_comm *rccl = nullptr;
  MapView view{};
};

template <typename MapType> using KmerIndex = Index<MapType, KmerParser<typename MapType::key_type>>;
template <typename MapType> using CountIndex = Index<MapType, KmerCountTupleParser<std::pair<typename MapType::key_type, typename MapType::mapped_type>>>;
template <typename MapType> using CountIndex2 = Index<MapType, KmerParser<typename MapType::key_type>>;
template <typename MapType> using PositionIndex = Index<MapType, KmerPositionTupleParser<std::pair<typename MapType::key_type, typename MapType::mapped_type>>>;
// PositionQualityIndex (kmer_index.hpp:405-406): mapped_type = std::pair<ShortSequenceKmerId, float>; on the device two 64-bit
// words per value (id, the float's bits in the low half of the second)
template <typename MapType> using PositionQualityIndex = Index<MapType, KmerPositionQualityTupleParser<std::pair<typename MapType::key_type, typename MapType::mapped_type>>>;

}  // namespace kmer
}  // namespace index

// ---------------------------------------------------------------------------
// bliss::io::KmerFileHelper::read_file_* (kmer_file_helper.hpp:550-633)
// ---------------------------------------------------------------------------
namespace io {
struct KmerFileHelper {
  template <typename KmerParser, template <typename> class SeqParser, template <typename, template <typename> class> class SeqIterType>
  static std::pair<size_t, size_t> read_file_posix(const std::string &filename, std::vector<typename KmerParser::value_type> &result,
                                                   const ::kmerind::comm &comm) {
    using Kmer = typename KmerParser::kmer_type;
    kmi_config c; std::memset(&c, 0, sizeof(c));
    c.k = Kmer::size; c.alphabet = Kmer::KmerAlphabet::KMI; c.seq_format = SeqParser<const unsigned char *>::KMI;
    c.index_kind = tuple_kind<typename KmerParser::value_type, Kmer>();
    c.seq_filter = SeqIterType<const unsigned char *, SeqParser>::KMI;
    kmi_ctx *ctx = nullptr;
    ::kmerind::check(nullptr, kmi_ctx_create(comm.device, comm.rank(), comm.size(), comm.stream, &ctx));
    kmi_tuples t{};
    kmi_status st = KMI_OK;
    if (comm.size() > 1) {
      // this rank's partition of the file: its byte range plus look-ahead, cut at record starts by the four-line rule (no
      // communication: both neighbours apply the rule at the same file position); the ids carry the file offsets
      if (c.seq_format == KMI_FMT_FASTA) {
        // FASTA: with a communicator (comm.transport or comm.unique_id) every rank reads its byte range plus look-ahead and the
        // blocks' summaries are gathered inside the library (collective: see Index::build_file); without one every rank reads
        // the file whole and keeps its block -- no communication at all
        kmi_comm *kc = nullptr;
        if (comm.transport.all_to_all_v) st = kmi_comm_create_transport(ctx, &comm.transport, &kc);
        else if (comm.unique_id.size() == KMI_COMM_ID_BYTES) st = kmi_comm_create(ctx, comm.unique_id.data(), &kc);
        if (st == KMI_OK && kc) {
          for (uint64_t look = 1ull << 16;; look *= 16) {
            ::bliss::index::kmer::detail::FileRange r = ::bliss::index::kmer::detail::read_file_range(filename, comm.rank(), comm.size(), look);
            int need_more = 0;
            st = kmi_extract_fasta_range_dist_host(ctx, &c, kc, r.bytes.data(), r.bytes.size(), r.offset, r.nominal, r.reaches_eof ? 1 : 0, r.prev_byte, &need_more, &t);
            if (st != KMI_OK || !need_more) break;
          }
          kmi_comm_destroy(kc);
        } else if (st == KMI_OK) {
          std::vector<uint8_t> whole = ::bliss::index::kmer::detail::read_whole_file(filename);
          st = kmi_extract_fasta_block_host(ctx, &c, whole.data(), whole.size(), (uint32_t)comm.rank(), (uint32_t)comm.size(), &t);
        }
      } else
      for (uint64_t look = 1ull << 20;; look *= 8) {
        ::bliss::index::kmer::detail::FileRange r = ::bliss::index::kmer::detail::read_file_range(filename, comm.rank(), comm.size(), look);
        int need_more = 0;
        st = kmi_extract_range_host(ctx, &c, r.bytes.data(), r.bytes.size(), r.offset, r.nominal, r.reaches_eof ? 1 : 0, &need_more, &t);
        if (st != KMI_OK || !need_more) break;
      }
    } else {
      std::vector<uint8_t> bytes = ::bliss::index::kmer::detail::read_whole_file(filename);
      st = kmi_extract_host(ctx, &c, bytes.data(), bytes.size(), 0, &t);
    }
    if (st != KMI_OK) { std::string m = kmi_last_error(ctx); kmi_ctx_destroy(ctx); if (st == KMI_ERR_PARSE) throw std::logic_error(m); throw std::invalid_argument(m); }
    const size_t before = result.size();
    result.reserve(before + t.n_tuples);
    for (uint64_t i = 0; i < t.n_tuples; ++i)
      result.push_back(make_value<typename KmerParser::value_type, Kmer>(t.kmers + i * Kmer::nWords, t.ids ? t.ids[i] : 1, t.quals ? t.quals[i] : 0.f));
    std::pair<size_t, size_t> r((size_t)t.n_seqs, (size_t)t.n_tuples);
    kmi_tuples_free(&t);
    kmi_ctx_destroy(ctx);
    return r;
  }
  template <typename KmerParser, template <typename> class SeqParser, template <typename, template <typename> class> class SeqIterType>
  static std::pair<size_t, size_t> read_file_mmap(const std::string &filename, std::vector<typename KmerParser::value_type> &result,
                                                  const ::kmerind::comm &comm) {
    return read_file_posix<KmerParser, SeqParser, SeqIterType>(filename, result, comm);
  }

 private:
  template <typename V, typename Kmer> static typename std::enable_if<std::is_same<V, Kmer>::value, uint32_t>::type tuple_kind() { return KMI_INDEX_COUNT; }
  template <typename V, typename Kmer> static typename std::enable_if<!std::is_same<V, Kmer>::value, uint32_t>::type tuple_kind() {
    return std::is_arithmetic<typename V::second_type>::value ? KMI_INDEX_COUNT :
           (sizeof(typename V::second_type) == sizeof(uint64_t) ? KMI_INDEX_POSITION : KMI_INDEX_POSQUAL);
  }
  template <typename V, typename Kmer> static typename std::enable_if<std::is_same<V, Kmer>::value, V>::type make_value(const uint64_t *w, uint64_t, float) { return Kmer(w); }
  // KmerCountTupleParser zips the k-mer with a constant 1 (kmer_parser.hpp:1008-1081), KmerPositionTupleParser with its id,
  // KmerPositionQualityTupleParser with (id, quality) (kmer_parser.hpp:577-900)
  template <typename V, typename Kmer> static typename std::enable_if<!std::is_same<V, Kmer>::value, V>::type make_value(const uint64_t *w, uint64_t v, float q) {
    uint32_t qb; std::memcpy(&qb, &q, 4);
    const uint64_t words[2] = {v, (uint64_t)qb};
    return V(Kmer(w), second_of<typename V::second_type>(words));
  }
  template <typename S> static typename std::enable_if<std::is_arithmetic<S>::value, S>::type second_of(const uint64_t *w) { return (S)w[0]; }
  template <typename S> static typename std::enable_if<!std::is_arithmetic<S>::value, S>::type second_of(const uint64_t *w) {
    return ::bliss::index::kmer::detail::value_words<S>::from(w);
  }
};
}  // namespace io
}  // namespace bliss

#endif  // KMERIND_KMER_INDEX_HPP
