"""What HBM sustains for pure writes, pure reads and copies on this box (torch fill_/sum/copy_ of 512 MB and 2 GB):
the ceiling for a launch whose traffic is almost all stores.   python tests/tools/hbm_write_rate.py"""
import torch


def timed(fn, n=20):
    for _ in range(3):
        fn()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2] * 1e-3


def main():
    dev = torch.device("cuda", 0)
    for mb in (512, 2048):
        n = mb * 1024 * 1024 // 4
        x = torch.empty(n, device=dev); y = torch.empty(n, device=dev)
        t = timed(lambda: x.fill_(1.5))
        print(f"{mb:5d} MB  fill_  {t * 1e6:8.1f} us  {n * 4 / t / 1e12:6.2f} TB/s written")
        t = timed(lambda: y.copy_(x))
        print(f"{mb:5d} MB  copy_  {t * 1e6:8.1f} us  {n * 4 / t / 1e12:6.2f} TB/s read + {n * 4 / t / 1e12:6.2f} TB/s written")
        t = timed(lambda: x.sum())
        print(f"{mb:5d} MB  sum    {t * 1e6:8.1f} us  {n * 4 / t / 1e12:6.2f} TB/s read")


if __name__ == "__main__":
    main()
