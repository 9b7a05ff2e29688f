// Test driver for the C++ shim classes (include/cpt_modules.hpp): reads a flat dump of cpt::Inputs written by
// tests/test_gpu_host_shim.py, builds PerturbationsModule and TransferModule exactly like the reference's Cosmology
// getters would (source/cosmology.cpp:30-35, 68-73), and writes the public tables back for comparison.
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <memory>
#include <vector>

#include "../include/cpt_modules.hpp"

static void rd(FILE* f, void* p, size_t n) {
  if (fread(p, 1, n, f) != n) { fprintf(stderr, "short read\n"); exit(2); }
}

int main(int argc, char** argv) {
  if (argc < 3) { fprintf(stderr, "usage: host_shim_demo <inputs.bin> <outputs.bin>\n"); return 2; }
  FILE* f = fopen(argv[1], "rb");
  if (!f) { perror("open"); return 2; }
  cpt::Inputs in;
  rd(f, &in.config, sizeof(in.config));
  rd(f, &in.tables, sizeof(in.tables));
  rd(f, &in.grid, sizeof(in.grid));
  const cpt_tables& t = in.tables;
  std::vector<double> tau(t.bt_size), bg((size_t)t.bt_size * t.bg_size), d2bg(bg.size()), z(t.tt_size),
      th((size_t)t.tt_size * t.th_size), d2th(th.size());
  rd(f, tau.data(), tau.size() * 8); rd(f, bg.data(), bg.size() * 8); rd(f, d2bg.data(), d2bg.size() * 8);
  rd(f, z.data(), z.size() * 8); rd(f, th.data(), th.size() * 8); rd(f, d2th.data(), d2th.size() * 8);
  // non-cold species: momentum grids q, w, dlnf0/dlnq of every species follow the tables
  std::vector<std::vector<double>> ncq(CPT_MAX_NCDM), ncw(CPT_MAX_NCDM), ncd(CPT_MAX_NCDM);
  if (in.config.has_ncdm)
    for (int n = 0; n < in.config.N_ncdm; n++) {
      const int nq = t.q_size_ncdm[n];
      ncq[n].resize(nq); ncw[n].resize(nq); ncd[n].resize(nq);
      rd(f, ncq[n].data(), nq * 8); rd(f, ncw[n].data(), nq * 8); rd(f, ncd[n].data(), nq * 8);
      in.tables.q_ncdm[n] = ncq[n].data(); in.tables.w_ncdm[n] = ncw[n].data(); in.tables.dlnf0_dlnq_ncdm[n] = ncd[n].data();
    }
  fclose(f);
  in.tables.tau_table = tau.data(); in.tables.background_table = bg.data(); in.tables.d2background_dtau2_table = d2bg.data();
  in.tables.z_table = z.data(); in.tables.thermodynamics_table = th.data(); in.tables.d2thermodynamics_dz2_table = d2th.data();
  int bad_flag = argc > 3 ? atoi(argv[3]) : 0;
  // flag 3: ignore the tables that came with the dump and recompute them on the host from the cosmological parameters
  // (cpt::HostTables); the parameter structs follow the arrays in the file
  std::unique_ptr<cpt::HostTables> host_tables;
  if (bad_flag == 3) {
    cpt_cosmo_params cosmo; cpt_thermo_params thermo;
    FILE* g = fopen(argv[1], "rb");
    fseek(g, -(long)(sizeof(cosmo) + sizeof(thermo)), SEEK_END);
    rd(g, &cosmo, sizeof(cosmo)); rd(g, &thermo, sizeof(thermo));
    fclose(g);
    std::fill(tau.begin(), tau.end(), 0.); std::fill(bg.begin(), bg.end(), 0.); std::fill(th.begin(), th.end(), 0.);   // (prove they are not used)
    host_tables = std::make_unique<cpt::HostTables>(cosmo, thermo);
    host_tables->fill(in);
  }
  // flag 4: the sharded constructors (cpt::Shard) with a communicator of ONE rank - everything a multi-GPU run executes
  // (cpt_comm_init, shard of the k loop, RCCL all-gather, shard of the multipoles, RCCL gather) on the one GPU a test box has
  char comm_id[CPT_COMM_ID_BYTES];
  if (bad_flag == 4) {
    if (cpt_comm_get_unique_id(comm_id)) { fprintf(stderr, "cpt_comm_get_unique_id failed\n"); return 11; }
    in.shard.rank = 0; in.shard.world = 1; in.shard.comm_id = comm_id;
  }
  // flag 5: two initial conditions in one module pair (adiabatic + cdm isocurvature): ic_size_[scalars] = 2
  if (bad_flag == 5) { in.n_ic = 2; in.ic[0] = CPT_IC_AD; in.ic[1] = CPT_IC_CDI; }
  // flag 6: modes = s,t in one module pair: md_size_ = 2; the tensor mode's config is the head of a second inputs file (argv[4])
  if (bad_flag == 6) {
    if (argc < 5) { fprintf(stderr, "flag 6 needs the tensor inputs file\n"); return 2; }
    FILE* g = fopen(argv[4], "rb");
    if (!g) { perror("open"); return 2; }
    rd(g, &in.config_tensors, sizeof(in.config_tensors));
    cpt_tables unused_tables; cpt_grid_params gt;
    rd(g, &unused_tables, sizeof(unused_tables)); rd(g, &gt, sizeof(gt));
    fclose(g);
    in.with_tensors = true;
    in.grid.l_tensor_max = gt.l_tensor_max;
  }
  if (bad_flag == 1) in.config.has_fld = 1;             // must raise std::invalid_argument
  if (bad_flag == 2) in.grid.k_step_transition = 0.;    // must raise std::invalid_argument (reference: class_test)
  try {
    auto pt = std::make_shared<const cpt::PerturbationsModule>(in);
    cpt::TransferModule tr(in, pt);
    FILE* o = fopen(argv[2], "wb");
    int md = pt->index_md_scalars_;
    int hdr[8] = {pt->k_size_[md], pt->k_size_cl_[md], pt->tau_size_, pt->tp_size_[md], tr.q_size_, tr.l_size_[md], tr.tt_size_[md], 0};
    fwrite(hdr, sizeof(int), 8, o);
    fwrite(pt->k_[md], 8, hdr[0], o);
    fwrite(pt->tau_sampling_, 8, hdr[2], o);
    for (int tp = 0; tp < hdr[3]; tp++) fwrite(pt->sources_[md][tp], 8, (size_t)hdr[2] * hdr[0], o);
    fwrite(tr.q_, 8, hdr[4], o);
    fwrite(tr.l_, 4, hdr[5], o);
    fwrite(tr.transfer_[md], 8, (size_t)hdr[6] * hdr[5] * hdr[4], o);
    if (bad_flag == 5) {   // the second initial condition: sources_[md][1 * tp_size + tp], transfer_[md] + 1 * tt_size * l_size * q_size
      if (pt->ic_size_[md] != 2 || pt->index_ic_ad_ != 0 || pt->index_ic_cdi_ != 1) { fprintf(stderr, "ic bookkeeping\n"); return 12; }
      for (int tp = 0; tp < hdr[3]; tp++) fwrite(pt->sources_[md][hdr[3] + tp], 8, (size_t)hdr[2] * hdr[0], o);
      fwrite(tr.transfer_[md] + (size_t)hdr[6] * hdr[5] * hdr[4], 8, (size_t)hdr[6] * hdr[5] * hdr[4], o);
    }
    if (bad_flag == 6) {   // the tensor mode
      if (pt->md_size_ != 2 || pt->index_md_tensors_ != 1 || tr.l_size_max_ != std::max(tr.l_size_[0], tr.l_size_[1])) { fprintf(stderr, "mode bookkeeping\n"); return 12; }
      const int mt = pt->index_md_tensors_;
      int hdr2[8] = {pt->k_size_[mt], pt->k_size_cl_[mt], pt->tp_size_[mt], tr.l_size_[mt], tr.tt_size_[mt], pt->ic_size_[mt], 0, 0};
      fwrite(hdr2, sizeof(int), 8, o);
      fwrite(pt->k_[mt], 8, hdr2[0], o);
      for (int tp = 0; tp < hdr2[2]; tp++) fwrite(pt->sources_[mt][tp], 8, (size_t)hdr[2] * hdr2[0], o);
      fwrite(tr.transfer_[mt], 8, (size_t)hdr2[4] * hdr2[3] * hdr[4], o);
    }
    fclose(o);
    printf("ok perturb %.2f ms transfer(LOS) %.3f ms\n", pt->kernel_ms(), tr.kernel_ms());
  } catch (std::invalid_argument& e) {
    printf("invalid_argument: %s\n", e.what());
    return 10;
  } catch (std::runtime_error& e) {
    printf("runtime_error: %s\n", e.what());
    return 11;
  }
  return 0;
}
