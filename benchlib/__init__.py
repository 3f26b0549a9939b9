"""bench.py in parts: `common` (timing, roofline, oracle, CPU baseline) and one module per BASELINE config."""
