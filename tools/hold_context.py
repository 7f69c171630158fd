import os, sys, time
sys.path.insert(0, os.getcwd())
import aardvark_amd
from aardvark_amd import synth, CompareConfig
contigs, batch = synth.config_genome(scale=0.05)
ctx = aardvark_amd.Context(0)
ctx.upload_reference(contigs)
rb = ctx.upload(batch)
ctx.compare_resident(rb, CompareConfig(enable_sequences=False))
ctx.synchronize()
print("holding", flush=True)
time.sleep(float(sys.argv[1]))
