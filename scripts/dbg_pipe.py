import torch, sys, math
sys.path.insert(0,".")
import flash_attention_dlrs_amd as fa
dev=torch.device("cuda:0")
torch.set_printoptions(linewidth=200, precision=3, sci_mode=False)
def ref(Q,K,V,causal):
    return torch.nn.functional.scaled_dot_product_attention(Q.float(),K.float(),V.float(),scale=1.0,is_causal=causal)
for shape in [(1,1,16,64),(1,1,64,64),(1,1,128,128),(1,1,321,128)]:
  for causal in (False,True):
    for var in ("mfma16p","mfma16p_w8"):
        torch.manual_seed(0)
        Q,K,V=(torch.randn(*shape,device=dev).bfloat16() for _ in range(3))
        O,L=fa.flash_attention_forward(Q,K,V,dev,causal=causal,variant=var)
        r=ref(Q,K,V,causal)
        S=Q.float()@K.float().transpose(-1,-2)
        if causal: S=S.masked_fill(~torch.ones_like(S,dtype=torch.bool).tril(),float("-inf"))
        lse=torch.logsumexp(S,-1)*math.log2(math.e)
        errL=(L.float().squeeze(-1)-lse).abs().max().item()
        err=(O.float()-r).abs()
        O1,_=fa.flash_attention_forward(Q,K,torch.ones_like(V),dev,causal=causal,variant=var)
        Vd=torch.arange(shape[3],device=dev).float().view(1,1,1,-1).expand(shape).contiguous().bfloat16()
        Od,_=fa.flash_attention_forward(Q,K,Vd,dev,causal=causal,variant=var)
        print(shape,causal,var,"errO %.3g errL %.3g"%(err.max().item(),errL),"V=1: maxdev %.3g"%(O1.float()-1).abs().max().item(),"V=d: maxdev %.3g"%(Od.float()-Vd.float()).abs().max().item(), "nan:",torch.isnan(O.float()).sum().item())
    if shape[2]==16:
        print("O[0,0,:4,:8]",O[0,0,:4,:8].float()); print("ref",r[0,0,:4,:8])
