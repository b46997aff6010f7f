import sys, time, torch
sys.path.insert(0, ".")
from frameino_amd.autoencoder_kl_wan import AutoencoderKLWan
from frameino_amd.configs import WAN22_VAE_CFG
vae = AutoencoderKLWan(**WAN22_VAE_CFG).random_init_(seed=7, device="cuda")
x = torch.rand(1, 3, 49, 704, 1280, device="cuda") * 2 - 1
for it in range(2):
    torch.cuda.synchronize(); t0=time.time(); vae.encode(x); torch.cuda.synchronize(); print("encode whole", round(time.time()-t0,4))
for n in (2,4,8):
    for i in (0, n//2):
        for it in range(2):
            torch.cuda.synchronize(); t0=time.time(); part,geo=vae.encode_slab(x,i,n); torch.cuda.synchronize(); t1=time.time()
        print(f"encode_slab {i} of {n}: {t1-t0:.4f} s")
part,geo=vae.encode_slab(x,0,1)
for it in range(2):
    torch.cuda.synchronize(); t0=time.time(); vae.encode_resume(part); torch.cuda.synchronize(); print("resume (replicated rest)", round(time.time()-t0,4))
