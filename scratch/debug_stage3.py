import copy, os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
from fullysparsefusion_amd import mmdet3d_plugin as plugin, synthetic, hip_ops
from fullysparsefusion_amd.compat import Config
from oracle import modules as omod
from oracle import scatter as osc
torch.manual_seed(0)
ROOT=os.path.join(os.path.dirname(__file__), '..')
cfg = Config.fromfile(os.path.join(ROOT,'configs/fsf_nuscenes.py'))
model = plugin.build_model(cfg.model).eval()
torch.nn.init.normal_(model.segmentor_updated_mlp[-1].weight, std=0.05)
for m in model.modules():
    if isinstance(m, torch.nn.BatchNorm1d):
        m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.5, 1.5); m.weight.data.uniform_(0.8, 1.2); m.bias.data.normal_(0, 0.1)
cpu = copy.deepcopy(model)
dev = torch.device('cuda:0'); model.to(dev)
f = synthetic.make_frame(1, 0)
pts8 = torch.from_numpy(f['points']); mask=torch.from_numpy(f['mask_data']); anno=torch.from_numpy(f['mask_anno']); L=torch.from_numpy(f['lidar2img'])
def d(a,b,name):
    a=a.detach().cpu().double(); b=b.detach().cpu().double()
    print(f'{name:28s} shape {tuple(a.shape)} maxabs {float((a-b).abs().max()):.3e} scale {float(b.abs().max()):.3e}')
with torch.no_grad():
    s1 = omod.fsf_stage1(cpu, pts8, mask, anno, L)
    cap={}
    orig=omod.sir_forward
    def capf(sir, points, feats, coors, f_cluster):
        cap['a']=(points,feats,coors,f_cluster); return orig(sir,points,feats,coors,f_cluster)
    omod.sir_forward=capf
    s3 = omod.fsf_stage3(cpu, s1)
    omod.sir_forward=orig
    p,fe,co,fc = cap['a']
    # GPU stage 3 with captured SIR inputs
    gcap={}
    gorig = model.backbone.forward
    def gcapf(points, features, coors, f_cluster=None):
        gcap['a']=(points,features,coors,f_cluster); return gorig(points,features,coors,f_cluster)
    model.backbone.forward = gcapf
    seg_dev = {k: s1[k].to(dev) for k in ["seg_points", "seg_logits", "seg_vote_preds", "offsets", "seg_feats", "batch_idx"]}
    l_feats, l_xyz, l_inds, _ = model.fsd_forward(seg_dev, [dict(lidar2img=f['lidar2img'])], run_head=False)
    model.backbone.forward = gorig
    gp,gfe,gco,gfc = gcap['a']
    d(gp,p,'sir in points'); d(gfe,fe,'sir in feats'); d(gfc,fc,'sir in f_cluster'); print('coors equal', bool((gco.cpu().long()==co.long()).all()))
    d(l_xyz, s3['cluster_xyz'], 'cluster_xyz')
    d(l_feats, s3['cluster_feats'], 'cluster_feats (full)')
    # SIR on identical (oracle) inputs
    pf, cf, oc = model.backbone(p.to(dev), fe.to(dev), co.to(dev), fc.to(dev))
    opf, ocf, ooc = orig(cpu.backbone, p, fe, co, fc)
    d(cf, ocf, 'SIR same-input cluster feats'); d(pf, opf, 'SIR same-input point feats')
    # block by block
    new_coors, unq_inv = torch.unique(co, return_inverse=True, dim=0)
    out_feats = fe; gout = fe.to(dev)
    from fullysparsefusion_amd.mmdet3d_plugin.ops.sst_ops import unique_with_plan
    gnc, ginv, _ = unique_with_plan(co.to(dev))
    print('inv equal', bool((ginv.cpu()==unq_inv).all()))
    for i,(blk, gblk) in enumerate(zip(cpu.backbone.block_list, model.backbone.block_list)):
        inf = torch.cat([p, out_feats],1); ginf = torch.cat([p.to(dev), gout],1)
        out_feats, grp = omod.sir_layer_forward(blk, inf, co, fc, unq_inv, new_coors)
        r = gblk(ginf, co.to(dev), fc.to(dev), return_both=True, unq_inv_once=ginv, new_coors_once=gnc)
        gout, ggrp = r[0], r[1]
        d(gout, out_feats, f'block{i} point feats'); d(ggrp, grp, f'block{i} group feats')
        # inside: rel_mlp
        xyz_norm = torch.tensor(blk.xyz_normalizer)
        x = torch.cat([inf[:, :3] / xyz_norm[None, :], inf[:, 3:]], dim=1)
        rel = blk.rel_mlp(fc / blk.rel_dist_scaler); grel = gblk.rel_mlp(fc.to(dev)/gblk.rel_dist_scaler)
        d(grel, rel, f'  block{i} rel_mlp')
        x = x*rel
        pf0 = blk.vfe_layers[0](x); gpf0 = gblk.vfe_layers[0](x.to(dev))
        d(gpf0, pf0, f'  block{i} vfe0 (same input)')
        lin = blk.vfe_layers[0].linear(x); glin = gblk.vfe_layers[0].linear(x.to(dev))
        d(glin, lin, f'  block{i} vfe0.linear'); 
        ln = blk.vfe_layers[0].norm(lin); gln = gblk.vfe_layers[0].norm(lin.to(dev))
        d(gln, ln, f'  block{i} vfe0.norm(same in)')
        print('   lin row var min', float(lin.var(1).min()), 'x absmax', float(x.abs().max()))
