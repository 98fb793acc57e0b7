"""Empty stand-in: /root/reference/Utils.py:3 imports librosa at module top, but nothing
on the Wave-U-Net hot path calls it.  TEST INFRASTRUCTURE ONLY."""
