"""Batched iteration over a ``StereoDataset(device="cuda")`` with the file parsing off the critical path.

The reference feeds its loops from ``torch.utils.data.DataLoader(..., num_workers=N, pin_memory=True)``
(adapt.py:224-231, train.py:230-240): worker PROCESSES produce finished host tensors.  The device path of this build
cannot live in worker processes (they would each need the GPU), and does not need to: its host half is pure file
parsing in PIL / NumPy, which releases the GIL, so a small pool of THREADS parses several samples at once while the
consumer uploads the raw samples and runs the decode kernels of the next batch on a side stream — one batch ahead of
the training / adaptation step that is running on the main stream.

Deterministic: the random crop / flip decisions are drawn by the consumer, in sample order, from Python's ``random``
(exactly the draws ``dataset[index]`` would make); the shuffle order comes from its own seeded generator."""
import collections
import random
from concurrent.futures import ThreadPoolExecutor

import torch


class DevicePrefetcher(object):
  def __init__(self, dataset, batch_size, shuffle=False, drop_last=False, num_threads=8, seed=0):
    if dataset.device is None:
      raise ValueError("DevicePrefetcher needs a StereoDataset(device=...): the host path goes through DataLoader workers")
    self.dataset, self.batch_size = dataset, int(batch_size)
    self.shuffle, self.drop_last = shuffle, drop_last
    self.num_threads = max(1, int(num_threads))
    self._epoch_rng = random.Random(seed)

  def __len__(self):
    n = len(self.dataset)
    return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

  def _batches(self):
    order = list(range(len(self.dataset)))
    if self.shuffle:
      self._epoch_rng.shuffle(order)
    for k in range(0, len(order), self.batch_size):
      idx = order[k:k + self.batch_size]
      if len(idx) == self.batch_size or not self.drop_last:
        yield idx

  def __iter__(self):
    ds = self.dataset
    main = torch.cuda.current_stream(ds.device)
    side = torch.cuda.Stream(ds.device)
    batches = list(self._batches())
    with ThreadPoolExecutor(max_workers=self.num_threads) as pool:
      parsed = collections.deque()                     # futures, two batches deep
      def submit(b):
        parsed.append([pool.submit(ds.parse, i) for i in batches[b]])
      for b in range(min(2, len(batches))):
        submit(b)
      ready = None                                     # (batch dict, event) decoded on the side stream
      for b in range(len(batches)):
        raws = [f.result() for f in parsed.popleft()]
        if b + 2 < len(batches):
          submit(b + 2)
        side.wait_stream(main)                         # buffers freed by the consumer are reused in order
        with torch.cuda.stream(side):
          samples = []
          for raw in raws:
            i, j = ds._window(raw[0].shape[0], raw[0].shape[1])
            flip = bool(ds.do_hflip and random.random() < 0.5)
            samples.append(ds.decode(raw, i, j, flip))
          batch = {k: torch.stack([s[k] for s in samples]) for k in samples[0]}
          done = torch.cuda.Event()
          done.record(side)
        if ready is not None:
          prev, ev = ready
          main.wait_event(ev)
          for t in prev.values():
            t.record_stream(main)
          yield prev
        ready = (batch, done)
      if ready is not None:
        prev, ev = ready
        main.wait_event(ev)
        for t in prev.values():
          t.record_stream(main)
        yield prev
