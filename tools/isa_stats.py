"""Instruction statistics of one kernel in a hipcc -S --offload-device-only assembly file:
   python3 tools/isa_stats.py /tmp/txh.s adc_smfmac_kernelILi32 [--dump out.s]"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
name = sys.argv[2]
m = re.search(r'^(_ZN\S*%s\S*):' % re.escape(name), s, re.M)
i = m.start()
j = s.index('.end_amdhsa_kernel', i)
body = s[i:j]
lines = body.split('\n')
cnt = collections.Counter()
for l in lines:
    mm = re.match(r'\s+([a-z_0-9]+)', l)
    if mm:
        cnt[mm.group(1)] += 1
print(m.group(1), len(lines), 'lines')
print('mfma', sum(v for k, v in cnt.items() if 'mfma' in k), 'scratch', sum(v for k, v in cnt.items() if k.startswith('scratch_')))
print(cnt.most_common(45))
if '--dump' in sys.argv:
    open(sys.argv[sys.argv.index('--dump') + 1], 'w').write(body)
