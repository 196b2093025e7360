// mvtn_amd/_mvr_torch_op.so — host-side plumbing only: the mesh step of MVRenderer.forward (models/renderer.py:79-113) as ONE torch
// operator with a C++ autograd node, so that an eager training step crosses from Python into native code twice (the forward
// call and `loss.backward()`) and the backward runs on the autograd engine's thread without ever taking the interpreter lock.
// the Python `torch.autograd.Function` of mvtn_amd/functional.py cost 85-178 us of host time per step
// (struct building, seven torch.empty, ctx bookkeeping, the engine calling back into Python) against 150 us of GPU time, so the
// eager C3 figure followed the host of the lease.
//
// No kernels and no arithmetic here: every call goes to the C ABI of include/mvr.h (libmvr_hip.so), through function addresses
// that mvtn_amd/_lib.py hands over — this file does not link against the library, and the library stays free of torch types.
// What the node does is what functional._RenderMeshViews does (same buffers, same saved tensors, same two library calls, same
// fall-back when somebody also differentiates through the returned cameras); tests/test_gpu_module.py compares the two bit for bit.
#include <ATen/ATen.h>
#include <torch/csrc/autograd/custom_function.h>
#include <torch/library.h>

#include <cstring>
#include <vector>

#include "../../include/mvr.h"

namespace {

using at::Tensor;
using torch::autograd::AutogradContext;
using torch::autograd::variable_list;

// The C ABI entry points (addresses handed over once by mvtn_amd/_lib.py: no link-time dependency on libmvr_hip.so).
struct Entry {
  decltype(&mvr_render_mesh_views_forward) fwd = nullptr;
  decltype(&mvr_render_mesh_views_backward) bwd = nullptr;
  decltype(&mvr_render_meshes_backward) bwd_images = nullptr;
  decltype(&mvr_look_at_backward) bwd_look_at = nullptr;
  decltype(&mvr_camera_position_backward) bwd_campos = nullptr;
  decltype(&mvr_mesh_workspace_bytes) ws_bytes = nullptr;
  decltype(&mvr_last_error) last_error = nullptr;
} g_entry;

void init(std::vector<int64_t> a) {
  TORCH_CHECK(a.size() == 7, "mvr::init: seven entry points expected");
  g_entry.fwd = reinterpret_cast<decltype(g_entry.fwd)>(a[0]);
  g_entry.bwd = reinterpret_cast<decltype(g_entry.bwd)>(a[1]);
  g_entry.bwd_images = reinterpret_cast<decltype(g_entry.bwd_images)>(a[2]);
  g_entry.bwd_look_at = reinterpret_cast<decltype(g_entry.bwd_look_at)>(a[3]);
  g_entry.bwd_campos = reinterpret_cast<decltype(g_entry.bwd_campos)>(a[4]);
  g_entry.ws_bytes = reinterpret_cast<decltype(g_entry.ws_bytes)>(a[5]);
  g_entry.last_error = reinterpret_cast<decltype(g_entry.last_error)>(a[6]);
}

void check(int rc, const char *what) {
  if (rc != MVR_OK) {
    const char *msg = g_entry.last_error ? g_entry.last_error() : nullptr;
    TORCH_CHECK(false, "mvr ", what, " failed (code ", rc, "): ", msg ? msg : "");
  }
}

// the packed meshes (mvtn_amd.structures.PackedMeshes) as the C ABI's view of them
mvr_mesh_batch mesh_batch(const Tensor &verts, const Tensor &faces, const Tensor &vert_off, const Tensor &face_off, const Tensor &vert_colors,
                          int64_t max_faces, int64_t max_verts) {
  mvr_mesh_batch mb;
  std::memset(&mb, 0, sizeof(mb));
  mb.n_meshes = (int32_t)(vert_off.numel() - 1);
  mb.total_verts = (int32_t)verts.size(0);
  mb.total_faces = (int32_t)faces.size(0);
  mb.max_faces = (int32_t)max_faces;
  mb.max_verts = (int32_t)max_verts;
  mb.verts = verts.data_ptr<float>();
  mb.faces = faces.data_ptr<int32_t>();
  mb.vert_off = vert_off.data_ptr<int32_t>();
  mb.face_off = face_off.data_ptr<int32_t>();
  mb.vert_colors = vert_colors.defined() ? vert_colors.data_ptr<float>() : nullptr;
  return mb;
}

struct RenderMeshViews : public torch::autograd::Function<RenderMeshViews> {
  // `params`: the bytes of an mvr_mesh_params on the host (a zero-copy view of the ctypes struct); `keep`: tensors whose
  // addresses that struct holds (the sampling window's draws), kept alive with the node.
  static variable_list forward(AutogradContext *ctx, const Tensor &azim, const Tensor &elev, const Tensor &dist,
                               const c10::optional<Tensor> &light_opt, const c10::optional<Tensor> &host_status, const Tensor &verts,
                               const Tensor &faces, const Tensor &vert_off, const Tensor &face_off, const c10::optional<Tensor> &vert_colors,
                               int64_t max_faces, int64_t max_verts, const Tensor &params, std::vector<Tensor> keep, int64_t max_trials, int64_t seed,
                               int64_t stream, bool need_grad) {
    TORCH_CHECK(g_entry.fwd, "mvr::init has not been called");
    TORCH_CHECK(params.device().is_cpu() && params.scalar_type() == at::kByte && params.is_contiguous() && (size_t)params.numel() == sizeof(mvr_mesh_params),
                "render_mesh_views: parameters of ", params.numel(), " bytes, mvr_mesh_params has ", sizeof(mvr_mesh_params));
    mvr_mesh_params p;
    std::memcpy(&p, params.data_ptr(), sizeof(p));
    if (need_grad) p.flags |= MVR_KEEP_PIXEL_LISTS;   // the rasteriser leaves its lists of covered pixels / winning faces for the backward
    const Tensor vcol = vert_colors.has_value() ? *vert_colors : Tensor();
    const mvr_mesh_batch mb = mesh_batch(verts, faces, vert_off, face_off, vcol, max_faces, max_verts);
    const int64_t B = mb.n_meshes, V = p.n_views, H = p.image_h, W = p.image_w, N = B * V;
    TORCH_CHECK(azim.numel() == N && elev.numel() == N && dist.numel() == N, "azim/elev/dist must have B*V = ", N, " elements");
    TORCH_CHECK(azim.is_cuda() && verts.is_cuda() && azim.scalar_type() == at::kFloat && elev.scalar_type() == at::kFloat && dist.scalar_type() == at::kFloat &&
                azim.is_contiguous() && elev.is_contiguous() && dist.is_contiguous(), "azim/elev/dist: contiguous float32 tensors on the GPU expected");
    const bool relative = !light_opt.has_value() || !light_opt->defined();
    const auto opt = azim.options();
    ctx->set_materialize_grads(false);   // unused outputs (R, T of the returned cameras) arrive undefined, not as zeros
    Tensor R = at::empty({N, 3, 3}, opt), T = at::empty({N, 3}, opt);
    Tensor status = at::empty({N + 2}, opt.dtype(at::kInt));
    Tensor aux = at::empty({5, N}, opt);   // rows 0-1: the angles used | rows 2-4 (as (N, 3)): the relative light
    float *aux_p = aux.data_ptr<float>();
    const int64_t ws_bytes = (int64_t)g_entry.ws_bytes(&mb, &p);
    Tensor ws = at::empty({ws_bytes}, opt.dtype(at::kByte));
    Tensor images = at::empty({B, V, 3, H, W}, opt);
    Tensor p2f = need_grad ? at::empty({N, H, W}, opt.dtype(at::kInt)) : Tensor();
    int32_t *st = status.data_ptr<int32_t>();
    int32_t *sp = (host_status.has_value() && host_status->defined()) ? host_status->data_ptr<int32_t>() : st;
    check(g_entry.fwd(&mb, &p, azim.data_ptr<float>(), elev.data_ptr<float>(), dist.data_ptr<float>(), (int32_t)max_trials, (uint64_t)seed,
                      relative ? nullptr : light_opt->data_ptr<float>(), R.data_ptr<float>(), T.data_ptr<float>(), aux_p, aux_p + N, aux_p + 2 * N, st + 2, sp,
                      images.data_ptr<float>(), p2f.defined() ? p2f.data_ptr<int32_t>() : nullptr, ws.data_ptr(), (size_t)ws_bytes,
                      reinterpret_cast<mvr_stream_t>(stream)),
          "mvr_render_mesh_views_forward");
    if (need_grad) {
      ctx->save_for_backward({azim, elev, dist, aux, R, T, relative ? aux : *light_opt, p2f, ws, verts, faces, vert_off, face_off, vcol});
      p.flags |= MVR_WORKSPACE_FROM_FORWARD;   // the backward reuses the vertex attributes the forward left in the workspace
      ctx->saved_data["params"] = std::string(reinterpret_cast<const char *>(&p), sizeof(p));
      ctx->saved_data["keep"] = keep;
      ctx->saved_data["relative"] = relative;
      ctx->saved_data["stream"] = stream;
      ctx->saved_data["max_faces"] = max_faces;
      ctx->saved_data["max_verts"] = max_verts;
    }
    ctx->mark_non_differentiable({status});
    return {images, R, T, status};
  }

  static variable_list backward(AutogradContext *ctx, variable_list grads_out) {
    const auto saved = ctx->get_saved_variables();
    // forward() saves only when the caller asked for the gradient of azim / elev / dist (need_grad).  A node also exists when some
    // OTHER tensor argument requires grad (light, verts, vert_colors, keep): none of those is differentiated here.
    TORCH_CHECK(saved.size() == 14, "mvr::render_mesh_views: backward reached a node whose forward ran with need_grad = false — only azim, "
                "elev and dist are differentiable (a light / vertex / colour tensor that requires grad gets no gradient from this operator)");
    const Tensor &azim = saved[0], &elev = saved[1], &dist = saved[2], &aux = saved[3], &R = saved[4], &T = saved[5], &light = saved[6], &p2f = saved[7], &ws = saved[8];
    const mvr_mesh_batch mb = mesh_batch(saved[9], saved[10], saved[11], saved[12], saved[13], ctx->saved_data["max_faces"].toInt(), ctx->saved_data["max_verts"].toInt());
    mvr_mesh_params p;
    std::memcpy(&p, ctx->saved_data["params"].toStringRef().data(), sizeof(p));
    const bool relative = ctx->saved_data["relative"].toBool();
    const mvr_stream_t stream = reinterpret_cast<mvr_stream_t>(ctx->saved_data["stream"].toInt());
    const int64_t N = R.size(0);
    const auto opt = R.options();
    float *aux_p = aux.data_ptr<float>();
    const float *light_p = relative ? aux_p + 2 * N : light.data_ptr<float>();
    const Tensor &g = grads_out[0], &gR_ext = grads_out[1], &gT_ext = grads_out[2];
    Tensor grads = at::empty({3, N}, opt);
    float *gp = grads.data_ptr<float>();
    if (g.defined() && !gR_ext.defined() && !gT_ext.defined()) {   // the training step: everything in one call
      const Tensor gc = g.scalar_type() == at::kFloat ? g.contiguous() : g.to(at::kFloat).contiguous();
      check(g_entry.bwd(&mb, &p, azim.data_ptr<float>(), elev.data_ptr<float>(), dist.data_ptr<float>(), aux_p, aux_p + N, R.data_ptr<float>(), T.data_ptr<float>(),
                        light_p, relative ? 1 : 0, p2f.data_ptr<int32_t>(), gc.data_ptr<float>(), gp, gp + N, gp + 2 * N, ws.data_ptr(), (size_t)ws.numel(), stream),
            "mvr_render_mesh_views_backward");
    } else {   // somebody also differentiated through the returned cameras (or only through them): chain the three backwards
      Tensor gR_c = at::zeros({N, 3, 3}, opt), gT_c = at::zeros({N, 3}, opt);
      Tensor gL = relative ? at::zeros({N, 3}, opt) : Tensor();
      if (g.defined()) {
        const Tensor gc = g.scalar_type() == at::kFloat ? g.contiguous() : g.to(at::kFloat).contiguous();
        check(g_entry.bwd_images(&mb, &p, R.data_ptr<float>(), T.data_ptr<float>(), light_p, p2f.data_ptr<int32_t>(), gc.data_ptr<float>(), gR_c.data_ptr<float>(),
                                 gT_c.data_ptr<float>(), gL.defined() ? gL.data_ptr<float>() : nullptr, ws.data_ptr(), (size_t)ws.numel(), stream),
              "mvr_render_meshes_backward");
      }
      if (gR_ext.defined()) gR_c = (gR_c + gR_ext).contiguous();
      if (gT_ext.defined()) gT_c = (gT_c + gT_ext).contiguous();
      check(g_entry.bwd_look_at(aux_p, aux_p + N, dist.data_ptr<float>(), (int32_t)N, gR_c.data_ptr<float>(), gT_c.data_ptr<float>(), gp, gp + N, gp + 2 * N, stream),
            "mvr_look_at_backward");
      if (gL.defined()) {
        Tensor g2 = at::empty({3, N}, opt);
        float *g2p = g2.data_ptr<float>();
        check(g_entry.bwd_campos(azim.data_ptr<float>(), elev.data_ptr<float>(), dist.data_ptr<float>(), (int32_t)N, gL.data_ptr<float>(), g2p, g2p + N, g2p + 2 * N, stream),
              "mvr_camera_position_backward");
        grads += g2;
      }
    }
    variable_list out(18);   // one slot per argument of forward(); only the three leaves carry a gradient
    out[0] = grads.select(0, 0); out[1] = grads.select(0, 1); out[2] = grads.select(0, 2);
    return out;
  }
};

std::vector<Tensor> render_mesh_views(const Tensor &azim, const Tensor &elev, const Tensor &dist, const c10::optional<Tensor> &light,
                                      const c10::optional<Tensor> &host_status, const Tensor &verts, const Tensor &faces, const Tensor &vert_off,
                                      const Tensor &face_off, const c10::optional<Tensor> &vert_colors, int64_t max_faces, int64_t max_verts,
                                      const Tensor &params, std::vector<Tensor> keep, int64_t max_trials, int64_t seed, int64_t stream, bool need_grad) {
  return RenderMeshViews::apply(azim, elev, dist, light, host_status, verts, faces, vert_off, face_off, vert_colors, max_faces, max_verts, params,
                                std::move(keep), max_trials, seed, stream, need_grad);
}

}  // namespace

TORCH_LIBRARY(mvr, m) {
  m.def("init(int[] entry_points) -> ()", init);
  m.def("render_mesh_views(Tensor azim, Tensor elev, Tensor dist, Tensor? light, Tensor? host_status, Tensor verts, Tensor faces, Tensor vert_off, "
        "Tensor face_off, Tensor? vert_colors, int max_faces, int max_verts, Tensor params, Tensor[] keep, int max_trials, int seed, int stream, "
        "bool need_grad) -> Tensor[]");
}

// (the autograd node is built inside: the operator is registered for every backend key at once, as a composite)
TORCH_LIBRARY_IMPL(mvr, CompositeImplicitAutograd, m) { m.impl("render_mesh_views", render_mesh_views); }
