import json,sys
for f in sys.argv[1:]:
    for l in open(f):
        if l.startswith("{"):
            o=json.loads(l); print(f.split('/')[-1], round(o["value"]/1e6,1), "Mtriples/s  step ms", round(o["ms_per_step"],3), "kernel ms", round(o["roofline"]["kernel_ms_avg"],3), "frac", round(o["roofline"]["frac"],3))
