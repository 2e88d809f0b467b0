// kmi_synth.cpp -- deterministic synthetic FASTQ of SURVEY.md 8(d) (host code).
//
// Counter-based so any read range can be produced independently and in parallel:
//   genome base i   = 2 bits of splitmix64(seed*A + i/32)            (i.i.d. uniform ACGT)
//   read r          = genome[start, start+L), start = mix(r,0) % (G-L+1),
//                     reverse-complemented iff mix(r,1) & 1, no errors, no N
//   qualities       = '#'..'I' (Phred 2..40) i.i.d. from mix(r, 2+j/8)
//   record          = '@' + 9-digit zero padded r + '\n' + bases + "\n+\n" + quals + '\n'
// (315 bytes for L = 150). Used by bench.py and the tests; the same bytes feed the GPU
// path and the CPU oracle.
#include <pthread.h>
#include <stdint.h>
#include <string.h>

#include <vector>

#include "../../include/kmerind_hip.h"

namespace {

inline uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

struct Gen {
  uint64_t seed, genome_len; uint32_t read_len;
  uint64_t gseed() const { return splitmix64(seed ^ 0x67656e6f6d65ull); }
  uint64_t rseed() const { return splitmix64(seed ^ 0x7265616473ull); }
  inline uint32_t base(uint64_t i, uint64_t gs) const { return (uint32_t)(splitmix64(gs + (i >> 5)) >> (2 * (i & 31))) & 3u; }
};

struct Job { Gen g; uint64_t first, n; uint8_t *out; };

void *worker(void *vp) {
  Job *j = (Job *)vp;
  const Gen &g = j->g;
  const uint64_t gs = g.gseed(), rs = g.rseed();
  const uint32_t L = g.read_len;
  const size_t rec = kmi_synth_fastq_bytes(1, L);
  static const char ACGT[4] = {'A', 'C', 'G', 'T'};
  for (uint64_t t = 0; t < j->n; ++t) {
    const uint64_t r = j->first + t;
    uint8_t *p = j->out + t * rec;
    *p++ = '@';
    uint64_t v = r % 1000000000ull;
    for (int d = 8; d >= 0; --d) { p[d] = (uint8_t)('0' + v % 10); v /= 10; }
    p += 9; *p++ = '\n';
    const uint64_t start = splitmix64(rs + 64 * r) % (g.genome_len - L + 1);
    const bool rc = splitmix64(rs + 64 * r + 1) & 1;
    {
      // walk the genome words once (32 bases per splitmix64 call)
      uint64_t gi = start, word = splitmix64(gs + (gi >> 5));
      for (uint32_t i = 0; i < L; ++i, ++gi) {
        if ((gi & 31) == 0 && i) word = splitmix64(gs + (gi >> 5));
        const uint32_t c = (uint32_t)(word >> (2 * (gi & 31))) & 3u;
        if (!rc) p[i] = (uint8_t)ACGT[c]; else p[L - 1 - i] = (uint8_t)ACGT[3u - c];
      }
    }
    p += L; *p++ = '\n'; *p++ = '+'; *p++ = '\n';
    for (uint32_t i = 0; i < L; i += 8) {
      uint64_t q = splitmix64(rs + 64 * r + 2 + (i >> 3));
      for (uint32_t b = 0; b < 8 && i + b < L; ++b) p[i + b] = (uint8_t)('#' + ((q >> (8 * b)) & 0xff) % 39u);
    }
    p += L; *p++ = '\n';
  }
  return nullptr;
}

}  // namespace

extern "C" size_t kmi_synth_fastq_bytes(uint64_t n_reads, uint32_t read_len) {
  return (size_t)n_reads * (size_t)(1 + 9 + 1 + read_len + 3 + read_len + 1);
}

extern "C" kmi_status kmi_synth_fastq(uint64_t seed, uint64_t genome_len, uint32_t read_len, uint64_t first_read,
                                      uint64_t n_reads, uint8_t *out, size_t out_capacity, uint32_t threads) {
  if (!out || read_len == 0 || genome_len < read_len || threads == 0) return KMI_ERR_INVALID;
  if (out_capacity < kmi_synth_fastq_bytes(n_reads, read_len)) return KMI_ERR_INVALID;
  const size_t rec = kmi_synth_fastq_bytes(1, read_len);
  std::vector<Job> jobs(threads);
  std::vector<pthread_t> th(threads);
  const uint64_t per = (n_reads + threads - 1) / threads;
  uint32_t started = 0;
  for (uint32_t t = 0; t < threads; ++t) {
    uint64_t f = (uint64_t)t * per;
    if (f >= n_reads) break;
    uint64_t n = (f + per <= n_reads) ? per : n_reads - f;
    jobs[t] = Job{Gen{seed, genome_len, read_len}, first_read + f, n, out + f * rec};
    pthread_create(&th[t], nullptr, worker, &jobs[t]);
    ++started;
  }
  for (uint32_t t = 0; t < started; ++t) pthread_join(th[t], nullptr);
  return KMI_OK;
}
