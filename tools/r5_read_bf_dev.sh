mkdir -p /tmp/rb && HPMN_DET_SCATTER=1 HPMN_READ_BF16=1 python tests/read_bf_worker.py /tmp/rb/a.npz && HPMN_DET_SCATTER=1 HPMN_READ_BF16=0 python tests/read_bf_worker.py /tmp/rb/b.npz && python - <<'P'
import numpy as np
a,b=np.load('/tmp/rb/a.npz'),np.load('/tmp/rb/b.npz')
for t in ('xlong','small'):
    ga,gb=a[t+'_grad'],b[t+'_grad']
    print(t,'pred',np.abs(a[t+'_pred']-b[t+'_pred']).max(),'ce rel',abs(a[t+'_ce']-b[t+'_ce'])/abs(b[t+'_ce']),'grad',np.abs(ga-gb).max()/np.abs(gb).max(), 'gmax', np.abs(gb).max(), 'tab', abs(a[t+'_table_grad_abs']-b[t+'_table_grad_abs'])/b[t+'_table_grad_abs'], 'pred range', b[t+'_pred'].min(), b[t+'_pred'].max())
P
