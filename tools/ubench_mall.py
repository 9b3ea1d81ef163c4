"""Does the Infinity Cache keep what a kernel has just WRITTEN for the next kernel's reads?  write N MB (fill), then read it (sum): effective read bandwidth by size;
and the same with a streaming read of another buffer of the same size in between (the traffic a transposition kernel's own reads would add)."""
import torch, time
dev = torch.device("cuda", 0)
def t(fn, reps=20):
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / reps
for mb in (32, 64, 96, 128, 176, 224, 256, 384, 512, 1024, 4096):
    n = mb * (1 << 20) // 4
    x = torch.empty(n, dtype=torch.float32, device=dev); y = torch.empty(n, dtype=torch.float32, device=dev); z = torch.empty(n, dtype=torch.float32, device=dev)
    y.fill_(2.0)
    def wr(): x.fill_(1.0)
    def rd(): return x.sum()
    def copy(): z.copy_(y)
    tw = t(wr); tr = t(rd)
    def wr_rd(): x.fill_(1.0); x.sum()
    def wr_copy_rd(): x.fill_(1.0); z.copy_(y); x.sum()
    twr = t(wr_rd); tc = t(copy); twcr = t(wr_copy_rd)
    print(f"{mb:5d} MB  write {mb/1e3/tw*1e3/1e3:6.2f} TB/s  read(repeated) {mb/1e3/tr:6.2f} TB/s  read-after-write {mb/1e3/max(twr-tw,1e-6):6.2f} TB/s  read-after-write+copy {mb/1e3/max(twcr-tw-tc,1e-6):6.2f} TB/s", flush=True)
