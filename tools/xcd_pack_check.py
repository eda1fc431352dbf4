"""us per iteration of 2..64-workgroup launches with the working groups packed on 1..7 XCDs (option xcd_pack)."""
import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
import torch
from gato_python_amd import synth
from gato_python_amd.solver import Solver


def main():
    for (S,C,K,dt,thr) in [(14,7,512,np.float32,512),(14,7,50,np.float64,256),(14,7,2000,np.float32,512),(14,7,4096,np.float32,512),(14,7,4096,np.float64,512),(32,16,1024,np.float32,512)]:
        s = synth.make_system(S,C,K,seed=0)
        for pack in (0,-1,2,4,6):
            sol = Solver(S,C,K,dt); sol.set_option("pcg_threads", thr); sol.set_option("xcd_pack", pack); sol.set_option("no_single_lds",1)
            dev = sol.upload_system(s); lam, dz = sol.new(S*K), sol.new(sol.N)
            sol.linsys(*dev, 0.0, 100, s.rho, lam, dz); torch.cuda.synchronize(); sol.check_status()
            sol.set_option("time_pcg",1); b=[sol.buffer_ptr(i) for i in (3,4,5)]; ms=[]
            for i in range(10):
                sol.pcg(b[0],b[1],b[2],0.0,100,lam=lam,check=False); ms.append(sol.pcg_last_ms())
            print(S,K,np.dtype(dt).name,"W",sol.get_option("last_groups"),"pack",pack,"us/iter %.3f"%(1e3*np.median(ms[2:])/100), flush=True)
            sol.close()


if __name__ == "__main__":
    main()
