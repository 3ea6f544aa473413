"""DESIGN.md and README.md are these templates with the figures of one bench line filled in:
   python tools/docs/fill_design.py profiles/<tag>_bench_default.json <tag>"""
import json,os,sys
HERE=os.path.dirname(os.path.abspath(__file__)); ROOT=os.path.dirname(os.path.dirname(HERE))
tmpl=open(os.path.join(HERE,'DESIGN.template.md')).read()
bench=sys.argv[1]; tag=sys.argv[2]
d=json.loads(open(bench).read().strip().splitlines()[-1]); c=d['config']; r=d['roofline']; k=c.get('k22_stress') or {}; cpu=d.get('cpu_baseline') or {}
rep={'@@VALUE@@':"%.1f"%d['value'],'@@VSAMPLES@@':" / ".join("%.1f"%v for v in c['value_samples']),'@@RESIDENT@@':"%.1f"%c['resident_proofs_per_s'],
 '@@RATIO@@':"%.3f"%c['streamed_over_resident'],'@@MS@@':"%.2f"%d['ms_per_step'],'@@LAT@@':"%.2f"%c['single_proof_latency_ms'],'@@LATSER@@':"%.2f"%c['single_proof_latency_ms_serial_key'],
 '@@HOSTCPU@@':"%.3f"%c['host_cpu_s_per_proof'],'@@L1MS@@':"%.3f"%r['avg_launch_ms'],'@@FRAC@@':"%.4f"%r['frac'],
 '@@TRAFFIC@@':("%.0f MB raw (%.1f× the algorithmic 109.3 MB)"%(r['traffic']/1e6,r['traffic']/r['algorithmic_bytes_per_launch'])) if r.get('traffic') else "no PMC summary for this build",
 '@@CPUS@@':"%.2f"%cpu.get('seconds_per_proof',0),'@@K22@@':"%.2f / %.2f ms, %.3f / %.3f ms"%(k.get('msm_ms',0),k.get('msm_skewed_ms',0),k.get('ntt_ms',0),k.get('intt_ms',0)),'@@TAG@@':tag}
for a,b in rep.items(): tmpl=tmpl.replace(a,b)
assert '@@' not in tmpl
open(os.path.join(ROOT,'DESIGN.md'),'w').write(tmpl)
print("DESIGN.md written:", len(tmpl.splitlines()), "lines")
# README
rt=open(os.path.join(HERE,'README.template.md')).read()
rep2=dict(rep)
rep2['@@K22MSM@@']="%.2f"%k.get('msm_ms',0); rep2['@@K22NTT@@']="%.2f"%k.get('ntt_ms',0)
for a,b in rep2.items(): rt=rt.replace(a,b)
assert '@@' not in rt
open(os.path.join(ROOT,'README.md'),'w').write(rt)
print("README.md written")
