import ctypes as C, sys
import numpy as np, torch
import kmerind_amd as K
from kmerind_amd import _lib as L
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream(dev)
ctx = K.Context(device=0, rank=0, nranks=1, stream=stream.cuda_stream)
cfg = K.make_config(31, "DNA", strand="canonical")
n_reads = int(sys.argv[1]); genome = int(sys.argv[2]); nch = int(sys.argv[3]); world = int(sys.argv[4])
host = K.synth_fastq(3, genome, n_reads, 150)
nbytes = int(host.nbytes)
d_bytes = torch.from_numpy(host).to(dev)
rec = nbytes // n_reads
bounds = [((n_reads * c // nch) // 16 * 16) * rec for c in range(nch)] + [nbytes]
n_kmers = n_reads * 120
chunk_kmers = max(((bounds[c + 1] - bounds[c]) // rec) * 120 for c in range(nch))
d_send = [torch.empty((chunk_kmers + 64, 1), dtype=torch.int64, device=dev) for _ in range(2)]
d_recv = torch.empty((int(n_kmers * 1.25) + 4096, 1), dtype=torch.int64, device=dev)
counts = np.zeros(world, dtype=np.uint64)
nt, ns = C.c_uint64(), C.c_uint64()
idx = K.CountIndex(ctx, cfg)
pos = 0
for c in range(nch):
    send = d_send[c & 1]
    ctx.check(L.lib.kmi_extract_route_dev(ctx.h, C.byref(cfg), C.c_void_p(d_bytes.data_ptr() + bounds[c]), bounds[c + 1] - bounds[c], world,
                                          C.c_void_p(send.data_ptr()), send.shape[0], C.byref(nt), C.byref(ns), counts.ctypes.data_as(C.c_void_p)))
    print("chunk", c, "nt", nt.value, "ns", ns.value, "counts sum", int(counts.sum()), "min key", int(send[:nt.value].min()), "zeros", int((send[:nt.value] == 0).sum()))
    d_recv[pos:pos + nt.value].copy_(send[:nt.value])
    pos += nt.value
torch.cuda.synchronize()
idx.insert_device(d_recv.data_ptr(), pos)
print("pos", pos, "distinct", idx.local_size())
idx2 = K.CountIndex(ctx, cfg)
idx2.build_device(d_bytes.data_ptr(), nbytes)
print("fused build distinct", idx2.local_size())
