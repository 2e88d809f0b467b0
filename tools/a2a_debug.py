import os, torch, torch.distributed as dist
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=dev)
for n in (100_000_000, 134_217_727, 134_217_728, 134_217_729, 140_000_000, 268_435_456, 268_435_457):
    src = torch.arange(n, dtype=torch.int64, device=dev).reshape(n, 1) * 7 + 3
    dst = torch.zeros((n + 10, 1), dtype=torch.int64, device=dev)
    w = dist.all_to_all_single(dst[5:5 + n], src[:n], output_split_sizes=[n], input_split_sizes=[n], async_op=True)
    w.wait()
    torch.cuda.synchronize()
    ok = bool((dst[5:5 + n] == src).all())
    bad = int((dst[5:5 + n] != src).sum())
    print("n", n, "ok", ok, "mismatches", bad, flush=True)
    del src, dst
dist.destroy_process_group()
