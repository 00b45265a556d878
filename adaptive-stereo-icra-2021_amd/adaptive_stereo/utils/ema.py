"""Exponential moving average used to smooth the FCS (reference adaptive_stereo/utils/ema.py:13)."""


def online_ema(s_last, v_new, weight=0.999):
  """s_last*weight + (1-weight)*v_new; works on floats and on 0-d device tensors alike (no sync)."""
  return s_last * weight + (1 - weight) * v_new
