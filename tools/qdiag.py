import os, numpy as np
import kmerind_amd as K
from tests import oracle as orc
ctx=K.Context(0)
k=31
s=orc.kspec(k,orc.DNA)
cfg=K.make_config(k,"DNA",index_kind="posqual")
for name in ("test.small.fastq","natural.fastq"):
    data=open(os.path.join("tests/golden/data",name),"rb").read()
    ex=orc.extract(s,data,orc.FASTQ,want_ids=True,want_quals=True)
    kmers,ids,quals,nseq=ctx.read_file(cfg,data,with_ids=True,with_quals=True)
    a=quals.view(np.uint32).astype(np.int64); b=ex["quals"].view(np.uint32).astype(np.int64)
    d=np.abs(a-b)
    print(name, len(a), "mismatch", int((d>0).sum()), "max ulp", int(d.max()), "first idx", np.nonzero(d)[0][:10], quals[np.nonzero(d)[0][:5]], ex["quals"][np.nonzero(d)[0][:5]])
import sys
sys.path.insert(0,'tests')
from tests.test_gpu_quality import _fastq_with_quals
rng=np.random.default_rng(31)
cases=[("plain",_fastq_with_quals(rng,400)),("zeros",_fastq_with_quals(rng,300,zero_frac=0.02)),("fullrange",_fastq_with_quals(rng,200,lo=33,hi=129)),("synth",bytes(K.synth_fastq(seed=6,genome_len=50_000,n_reads=1500)))]
for name,data in cases:
    ex=orc.extract(s,data,orc.FASTQ,want_ids=True,want_quals=True)
    kmers,ids,quals,nseq=ctx.read_file(cfg,data,with_ids=True,with_quals=True)
    a=quals.view(np.uint32).astype(np.int64); b=ex["quals"].view(np.uint32).astype(np.int64)
    d=np.abs(a-b); nz=np.nonzero(d)[0]
    print(name, len(a), "mismatch", len(nz), "max ulp", int(d.max()) if len(d) else 0, nz[:8], quals[nz[:4]], ex["quals"][nz[:4]])
