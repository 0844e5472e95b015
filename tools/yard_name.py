import torch
a=torch.randn(32362,4352,device='cuda').bfloat16(); b=(torch.randn(12288,4352,device='cuda')*0.02).bfloat16()
for _ in range(3): torch.matmul(a,b.t())
a=torch.randn(8192,8192,device='cuda').bfloat16(); b=torch.randn(8192,8192,device='cuda').bfloat16()
for _ in range(3): torch.matmul(a,b.t())
torch.cuda.synchronize()
