import torch, hashlib, platform, subprocess
def h(t): return hashlib.md5(t.contiguous().numpy().tobytes()).hexdigest()[:10]
print(torch.__config__.show().split("\n")[0:3], torch.backends.cpu.get_cpu_capability())
try: print(subprocess.run("lscpu | grep -E 'Model name|Flags' | cut -c1-200", shell=True, capture_output=True, text=True).stdout)
except Exception as e: print(e)
R=64; pitch=3.74e-6
wl=torch.tensor([638e-9,520e-9,450e-9])
fx=torch.fft.fftfreq(R,pitch); print("fx",h(fx), fx[1].item().hex() if hasattr(float,'hex') else 0)
sq=fx**2; print("fx2",h(sq))
rho=fx.unsqueeze(1)**2+fx.unsqueeze(0)**2; print("rho2",h(rho))
l2=wl**2; print("wl2",h(l2), [float(x).hex() for x in l2])
inv=1/l2; print("inv",h(inv), [float(x).hex() for x in inv])
diff=inv.view(-1,1,1)-rho.unsqueeze(0); print("diff",h(diff))
w=torch.sqrt(torch.clamp(diff,min=0)); print("w",h(w))
d=torch.tensor([1e-3])
th=-2j*torch.pi*d.view(-1,1,1,1)*w; print("theta",h(torch.view_as_real(th)))
H=torch.exp(th); print("H",h(torch.view_as_real(H)))
u=torch.fft.fftfreq(R).unsqueeze(-1); v=torch.fft.fftfreq(R).unsqueeze(0); D=torch.sqrt(u**2+v**2)*R; print("D",h(D))
