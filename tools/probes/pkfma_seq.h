#define PK_PRO \
  "s_mov_b32 s10, 0\n" \
  "s_mov_b32 s11, 0\n" \
  "s_mov_b32 s3, 0\n" \
  "s_add_u32 s14, %[base], 16*(6+2)\n" \
  "s_add_u32 s15, %[base], 16*(6+3)\n" \
  "s_add_u32 s80, %[base], 16*(6+4)\n" \
  "s_add_u32 s2, %[base], 16*(6+5)\n" \
  "s_add_u32 s98, %[base], 16*(6+6)\n" \
  "s_add_u32 s99, %[base], 16*(6+7)\n" \
  "s_add_u32 s92, %[base], 16*(6+8)\n" \
  "s_add_u32 s93, %[base], 16*(6+9)\n" \
  "s_add_u32 s40, %[base], 16*(6+10)\n" \
  "s_add_u32 s41, %[base], 16*(6+11)\n" \
  "s_add_u32 s87, %[base], 16*18\n" \
  "s_add_u32 s76, %[base], 16*19\n" \
  "s_add_u32 s12, %[base], 16*(6+0)\n" \
  "s_add_u32 s13, %[base], 16*(6+1)\n" \
  "v_writelane_b32 v127, s12, 8\n" \
  "v_writelane_b32 v127, s13, 9\n" \
  "s_mov_b32 s12, -1\n" \
  "v_writelane_b32 v126, s12, 34\n" \
  "v_writelane_b32 v126, s12, 35\n" \
  "v_mov_b32 v26, %[x0]\n" \
  "v_mov_b32 v27, %[x1]\n" \
  "v_mov_b32 v28, %[x2]\n" \
  "v_mov_b32 v29, %[x3]\n" \
  "v_mov_b32 v30, %[x4]\n" \
  "v_mov_b32 v31, %[x5]\n" \
  "v_mov_b32 v32, %[x6]\n" \
  "v_mov_b32 v33, %[x7]\n" \
  "v_mov_b32 v34, %[x8]\n" \
  "v_mov_b32 v35, %[x9]\n" \
  "v_mov_b32 v36, %[x10]\n" \
  "v_mov_b32 v37, %[x11]\n"
#define PK_BODY \
  "v_readlane_b32 s12, v127, 8\n" \
  "s_nop 1\n" \
  "v_mov_b32_e32 v0, s12\n" \
  "v_readlane_b32 s12, v127, 9\n" \
  "ds_read_b128 v[38:41], v0\n" \
  "s_waitcnt lgkmcnt(0)\n" \
  "v_pk_fma_f32 v[38:39], v[38:39], v[26:27], 0 op_sel_hi:[1,0,0]\n" \
  "v_mov_b32_e32 v0, s12\n" \
  "ds_read_b128 v[42:45], v0\n" \
  "v_mov_b32_e32 v0, s14\n" \
  "ds_read_b128 v[46:49], v0\n" \
  "v_mov_b32_e32 v0, s15\n" \
  "v_readlane_b32 s12, v126, 34\n" \
  "s_waitcnt lgkmcnt(1)\n" \
  "v_pk_fma_f32 v[38:39], v[42:43], v[26:27], v[38:39] op_sel:[0,1,0]\n" \
  "v_readlane_b32 s13, v126, 35\n" \
  "s_waitcnt lgkmcnt(0)\n" \
  "v_pk_fma_f32 v[42:43], v[46:47], v[28:29], v[38:39] op_sel_hi:[1,0,1]\n" \
  "v_pk_fma_f32 v[38:39], v[40:41], v[26:27], 0 op_sel_hi:[1,0,0]\n" \
  "s_and_b64 vcc, exec, s[12:13]\n" \
  "v_pk_fma_f32 v[38:39], v[44:45], v[26:27], v[38:39] op_sel:[0,1,0]\n" \
  "s_nop 0\n" \
  "v_pk_fma_f32 v[46:47], v[48:49], v[28:29], v[38:39] op_sel_hi:[1,0,1]\n" \
  "ds_read_b128 v[38:41], v0\n" \
  "v_mov_b32_e32 v0, v29\n" \
  "s_waitcnt lgkmcnt(0)\n" \
  "v_pk_fma_f32 v[38:39], v[38:39], v[0:1], v[42:43] op_sel_hi:[1,0,1]\n" \
  "v_mov_b32_e32 v42, s80\n" \
  "ds_read_b128 v[42:45], v42\n" \
  "v_pk_fma_f32 v[40:41], v[40:41], v[0:1], v[46:47] op_sel_hi:[1,0,1]\n" \
  "s_waitcnt lgkmcnt(0)\n" \
  "v_pk_fma_f32 v[38:39], v[42:43], v[30:31], v[38:39] op_sel_hi:[1,0,1]\n" \
  "v_mov_b32_e32 v42, s2\n" \
  "ds_read_b128 v[48:51], v42\n" \
  "v_mov_b32_e32 v42, s98\n" \
  "ds_read_b128 v[52:55], v42\n" \
  "v_mov_b32_e32 v42, s99\n" \
  "ds_read_b128 v[56:59], v42\n" \
  "s_waitcnt lgkmcnt(2)\n" \
  "v_pk_fma_f32 v[38:39], v[48:49], v[30:31], v[38:39] op_sel:[0,1,0]\n" \
  "v_mov_b32_e32 v42, v33\n" \
  "s_waitcnt lgkmcnt(1)\n" \
  "v_pk_fma_f32 v[38:39], v[52:53], v[32:33], v[38:39] op_sel_hi:[1,0,1]\n" \
  "v_mov_b32_e32 v48, v37\n" \
  "s_waitcnt lgkmcnt(0)\n" \
  "v_pk_fma_f32 v[38:39], v[56:57], v[42:43], v[38:39] op_sel_hi:[1,0,1]\n" \
  "v_mov_b32_e32 v43, s92\n" \
  "ds_read_b128 v[60:63], v43\n" \
  "v_mov_b32_e32 v43, s93\n" \
  "ds_read_b128 v[90:93], v43\n" \
  "v_mov_b32_e32 v43, s40\n" \
  "ds_read_b128 v[94:97], v43\n" \
  "v_mov_b32_e32 v43, s41\n" \
  "ds_read_b128 v[120:123], v43\n" \
  "v_mov_b32_e32 v43, s87\n" \
  "ds_read_b128 v[72:75], v43\n" \
  "v_mov_b32_e32 v43, s76\n" \
  "ds_read_b128 v[68:71], v43\n" \
  "s_waitcnt lgkmcnt(5)\n" \
  "v_pk_fma_f32 v[38:39], v[60:61], v[34:35], v[38:39] op_sel_hi:[1,0,1]\n" \
  "v_pk_fma_f32 v[40:41], v[44:45], v[30:31], v[40:41] op_sel_hi:[1,0,1]\n" \
  "s_waitcnt lgkmcnt(4)\n" \
  "v_pk_fma_f32 v[38:39], v[90:91], v[34:35], v[38:39] op_sel:[0,1,0]\n" \
  "v_pk_fma_f32 v[40:41], v[50:51], v[30:31], v[40:41] op_sel:[0,1,0]\n" \
  "s_waitcnt lgkmcnt(3)\n" \
  "v_pk_fma_f32 v[38:39], v[94:95], v[36:37], v[38:39] op_sel_hi:[1,0,1]\n" \
  "v_pk_fma_f32 v[40:41], v[54:55], v[32:33], v[40:41] op_sel_hi:[1,0,1]\n" \
  "s_waitcnt lgkmcnt(2)\n" \
  "v_pk_fma_f32 v[38:39], v[120:121], v[48:49], v[38:39] op_sel_hi:[1,0,1]\n" \
  "s_waitcnt lgkmcnt(0)\n" \
  "v_pk_fma_f32 v[38:39], v[38:39], v[72:73], v[68:69]\n" \
  "s_nop 0\n" \
  "v_max_f32_e32 v43, 0, v38\n" \
  "v_pk_fma_f32 v[40:41], v[58:59], v[42:43], v[40:41] op_sel_hi:[1,0,1]\n" \
  "v_max_f32_e32 v49, 0, v39\n" \
  "v_pk_fma_f32 v[40:41], v[62:63], v[34:35], v[40:41] op_sel_hi:[1,0,1]\n" \
  "v_cndmask_b32_e64 v39, v49, v39, s[10:11]\n" \
  "v_pk_fma_f32 v[40:41], v[92:93], v[34:35], v[40:41] op_sel:[0,1,0]\n" \
  "v_cndmask_b32_e64 v38, v43, v38, s[10:11]\n" \
  "v_pk_fma_f32 v[40:41], v[96:97], v[36:37], v[40:41] op_sel_hi:[1,0,1]\n" \
  "s_nop 0\n" \
  "v_pk_fma_f32 v[40:41], v[122:123], v[48:49], v[40:41] op_sel_hi:[1,0,1]\n" \
  "s_nop 0\n" \
  "v_pk_fma_f32 v[40:41], v[40:41], v[74:75], v[70:71]\n" \
  "s_nop 0\n" \
  "v_max_f32_e32 v0, 0, v40\n" \
  "v_max_f32_e32 v42, 0, v41\n" \
  "v_cndmask_b32_e64 v40, v0, v40, s[10:11]\n" \
  "v_mov_b32_e32 v0, s3\n" \
  "v_cndmask_b32_e64 v41, v42, v41, s[10:11]\n"
#define PK_EPI \
  "v_mov_b32 %[u0], v38\n" \
  "v_mov_b32 %[u1], v39\n" \
  "v_mov_b32 %[u2], v40\n" \
  "v_mov_b32 %[u3], v41\n"
