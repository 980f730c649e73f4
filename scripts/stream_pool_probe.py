"""GPU: what the FIRST torch.cuda.Stream() of a process costs (torch creates its whole stream pool then: 3 priorities x 32 streams),
and whether another thread can run Python meanwhile (does the call hold the GIL?)."""
import threading, time
import torch
torch.zeros(1, device="cuda"); torch.cuda.synchronize()
ticks = []
stop = False


def spin():
    while not stop:
        ticks.append(time.perf_counter())


th = threading.Thread(target=spin, daemon=True)
th.start()
time.sleep(0.02)
t0 = time.perf_counter()
s = torch.cuda.Stream()
t1 = time.perf_counter()
s2 = torch.cuda.Stream()
t2 = time.perf_counter()
s3 = torch.cuda.Stream(priority=-1)
t3 = time.perf_counter()
stop = True
th.join()
gaps = [b - a for a, b in zip(ticks, ticks[1:]) if t0 <= a <= t3]
print("first Stream(): %.1f ms, second %.2f ms, first high-priority %.2f ms; longest stall of a Python thread meanwhile: %.1f ms"
      % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, max(gaps) * 1e3 if gaps else -1))
