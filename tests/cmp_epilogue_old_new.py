"""Round 6: the fused ReadOut tail's epilogue before / after experiments #21 (bias / multiplier table and W2 through LDS) on the four
fused-head shapes the long fuzz run flagged (3 - 14 of 25 k - 100 k outputs beyond the checker's tolerance) + two more: BIT-IDENTICAL
outputs, same error against the fp32 reference -- the flags were the checker's allowance (one flipped bf16 hidden unit moves all
20 outputs of its pixel), since widened in tests/fuzz_conv.py.  Needs the old kernel as a variant library:

    C=$(git log --format=%H --grep="Fused ReadOut tail epilogue" | tail -1)
    git show $C^:celldetection_amd/csrc/conv_igemm.hip > celldetection_amd/csrc/_old.hip
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c celldetection_amd/csrc/_old.hip -o celldetection_amd/build/variants/conv_igemm_oldepi.o
    hipcc --offload-arch=gfx950 -shared -fPIC celldetection_amd/build/variants/conv_igemm_oldepi.o \
          $(ls celldetection_amd/build/*.o | grep -v "/conv_igemm.o$" | grep -v "/conv_igemm_clock.o$") -o celldetection_amd/build/variants/libcpn_oldepi.so
"""
import os, sys, subprocess, json, torch
ROOT=os.environ.get('GRAFT_REPO_ROOT','/root/repo')
CASES=[{'k': 7, 'n': 1, 'seed': 291604789, 'h': 40, 'w': 32, 'cin': 256, 'cout': 8, 'act': 'relu', 'bias': True, 'bn': True, 'fuse_cout': 20, 'fuse_act': 'tanh_scaled'},
{'k': 7, 'n': 1, 'seed': 185335922, 'h': 32, 'w': 40, 'cin': 64, 'cout': 8, 'act': 'relu', 'bias': True, 'bn': False, 'fuse_cout': 20, 'fuse_act': 'tanh_scaled'},
{'k': 7, 'n': 3, 'seed': 1067583490, 'h': 72, 'w': 24, 'cin': 64, 'cout': 64, 'act': 'relu', 'bias': False, 'bn': True, 'fuse_cout': 20, 'fuse_act': 'tanh_scaled'},
{'k': 7, 'n': 1, 'seed': 49407645, 'h': 44, 'w': 32, 'cin': 24, 'cout': 64, 'act': 'relu', 'bias': False, 'bn': True, 'fuse_cout': 20, 'fuse_act': 'tanh_scaled'},
{'k': 7, 'n': 2, 'seed': 5, 'h': 64, 'w': 64, 'cin': 128, 'cout': 256, 'act': 'relu', 'bias': True, 'bn': True, 'fuse_cout': 20, 'fuse_act': 'none'},
{'k': 3, 'n': 2, 'seed': 6, 'h': 32, 'w': 64, 'cin': 64, 'cout': 128, 'act': 'relu', 'bias': True, 'bn': True, 'fuse_cout': 2, 'fuse_act': 'sigmoid'}]
if len(sys.argv)>1:
    sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
    from test_gpu_kernels import run_conv
    outs=[]
    for c in CASES:
        got,ref,f32=run_conv(torch.device('cuda:0'),**c); outs.append((got,ref))
    torch.save(outs, sys.argv[1]); sys.exit(0)
res={}
for tag,lib in (('new',None),('old',os.path.join(ROOT,'celldetection_amd/build/variants/libcpn_oldepi.so'))):
    env=dict(os.environ); env.pop('CPN_HIP_LIB',None)
    if lib: env['CPN_HIP_LIB']=lib
    subprocess.run([sys.executable,__file__,f'/tmp/epi_{tag}.pt'],env=env,check=True,capture_output=True)
    res[tag]=torch.load(f'/tmp/epi_{tag}.pt')
for i,c in enumerate(CASES):
    a,ref=res['new'][i]; b,_=res['old'][i]
    print(i, 'identical' if torch.equal(a,b) else f'DIFFER max {(a-b).abs().max().item():.3e} n {(a!=b).sum().item()}', 'err new', f'{(a-ref).abs().max().item():.3e}', 'err old', f'{(b-ref).abs().max().item():.3e}')
