"""Online validation set buffer: reservoir sampling (Algorithm R).

Reference: adaptive_stereo/utils/stereo_reservoir.py.  Host-side control logic; the entries are
whatever the caller stores (device tensors for the images, a float or 0-d tensor for the value).
"""
import random


class StereoReservoir(object):
  def __init__(self, max_size, min_heap=True):
    self.max_size = max_size
    self.buf = []
    self.indices = set()
    self.i = 0          # number of offers so far

  def add(self, img_l, img_r, value, img_index):
    """Offers an item; returns True when it was stored.  An index already present is refused."""
    self.i += 1
    if img_index in self.indices:
      return False
    entry = [value, img_index, img_l, img_r]
    if len(self.buf) < self.max_size:
      self.buf.append(entry)
      self.indices.add(img_index)
      return True
    slot = random.randint(1, self.i)
    if slot <= self.max_size:
      self.buf[slot - 1] = entry
      return True
    return None         # the reference falls off the end of the function here

  def update_value(self, buf_index, new_value):
    self.buf[buf_index][0] = new_value

  def size(self):
    return len(self.buf)

  def average_value(self):
    return sum(item[0] for item in self.buf) / len(self.buf)
