"""profiles/r01_pmc_fetch.csv + r01_pmc_write.csv -> profiles/pmc_traffic.json: fabric-side bytes per
launch for each kernel class of the DDPM step (read by bench.py for roofline.traffic).

Correction (MI355X_MICROARCH.md, "HBM [CDNA4]"): on gfx950 FETCH_SIZE reports exactly half of the
bytes of wide (16 B/lane) coalesced reads, global_load and LDS-DMA alike -> doubled here.  WRITE_SIZE
is taken as is (the stores of these kernels are 8-16 B/lane row-major; it matches the algorithmic
output bytes of every GEMM to within 5 %).  Both counters are in KiB and include Infinity-Cache hits.
Usage: python tools/pmc_traffic.py [tag]"""
import csv, json, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else 'r01'
CLASS = [  # (substring of the kernel name, class name of msd_profile_steps)
    ('EpiGeglu', 'gemm_mlp_in_geglu'), ('EpiQKV', 'gemm_qkv'),
    ('64, 32, 4, EpiResidualNorm', 'gemm_mlp_out'), ('32, 32, 4, EpiResidualNorm', 'gemm_attn_out+gemm_cross_out'),
    ('32, 32, 4, EpiStoreBf16', 'gemm_cross_q'), ('attention_kernel', 'attn_self+attn_cross'),
    ('attention_merge_kernel', 'attn_cross(merge)'), ('final_proj_f32_kernel', 'final_proj_f32'),
    ('EpiInProj', 'in_proj_f32'), ('sampler_step_kernel', 'sampler_step'),
]
def load(name):
  with open(os.path.join(root, 'profiles', '%s_pmc_%s.csv' % (tag, name))) as f:
    return {r['kernel']: r for r in csv.DictReader(f)}
fetch, write = load('fetch'), load('write')
out = {}
for k, r in fetch.items():
  for sub, cls in CLASS:
    if sub in k and not (sub == 'attention_kernel' and 'merge' in k):
      fk = float(r['FETCH_SIZE'])
      wk = float(write.get(k, {}).get('WRITE_SIZE', 0.0))
      out[cls] = {'kernel': k, 'dispatches': int(r['dispatches']), 'fetch_size_kib_raw': fk, 'write_size_kib': wk,
                  'bytes_per_launch': round((2.0 * fk + wk) * 1024)}
      break
json.dump({'source': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/prof_pmc.sh) over bench.py '
                     '--num-steps 24, base_with_context B=1 bf16x3; profiles/%s_pmc_fetch.csv, %s_pmc_write.csv' % (tag, tag),
           'correction': 'FETCH_SIZE doubled (gfx950 counts 16 B/lane reads at half size: MI355X_MICROARCH.md HBM section); '
                         'WRITE_SIZE as reported; counters include Infinity-Cache hits',
           'per_class': out}, open(os.path.join(root, 'profiles', 'pmc_traffic.json'), 'w'), indent=1)
print(json.dumps({k: v['bytes_per_launch'] for k, v in out.items()}, indent=1))
