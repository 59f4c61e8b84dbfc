import subprocess, time, os, sys
R=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
os.chdir("/dev/shm"); open("empty","wb").close(); open("one","wb").write(os.urandom(4<<20))
def t(cmd):
    s=time.perf_counter(); subprocess.run(cmd, capture_output=True); return time.perf_counter()-s
for name,cmd in (("hipmin",[R+"/tools/ubench/hipmin"]),("4mc empty",[R+"/4mc_amd/bin/4mc","-f","empty","e.4mc"]),("4mc one block",[R+"/4mc_amd/bin/4mc","-f","one","o.4mc"]),("4mc -d one block",[R+"/4mc_amd/bin/4mc","-d","-f","o.4mc","back"]),("ref one block",[R+"/oracle/_ref/4mc_ref","-f","one","r.4mc"])):
    print(name, " ".join("%.3f"%t(cmd) for _ in range(3)))
