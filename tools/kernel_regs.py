"""Print registers / spills / LDS of every kernel in an AMDGPU assembly file (hipcc -save-temps): tools/kernel_regs.py file.s [filter]"""
import re
import sys

txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ''
for blk in txt.split('  - .agpr_count:')[1:]:
    get = lambda k: (re.search(r'\.%s:\s+(\S+)' % k, blk) or [None, '?'])[1]
    name = get('name')
    if flt in name:
        print('%-90s vgpr %s agpr %s sgpr %s spill %s scratch %s lds %s' % (name[:90], get('vgpr_count'), blk.split()[0], get('sgpr_count'),
                                                                         get('vgpr_spill_count'), get('private_segment_fixed_size'),
                                                                         get('group_segment_fixed_size')))
