// node_api_bench.cc -- what a drop-in caller of the node API gets: the reference's canonical C++ graph (src/beamformerDS.cc:144-223:
// SampleFeature x N -> OverSampledDFTAnalysisBank x N -> SubbandGSC -> OverSampledDFTSynthesisBank) built on in-memory
// utterances and pulled block by block through next() until jiterator_error, exactly as the reference's main pulls it.
// Two forms: G graphs pulled one after the other (each an independent S = 1 launch sequence), and the same G graphs in a
// SubbandGraphPool (one S = G launch per round).  Prints one JSON line: frames/s of the whole run and where the host time went
// (common/devmem.h, btk_node_timers): pulling the sources, uploads, device work incl. waits, and the rest = serving next().
//
// usage: node_api_bench <coeffs.f64 (h then g, m*M doubles each)> <M> <m> <r> <channels> <frames per graph> <graphs> <block_frames> <pool 0|1>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <list>
#include <vector>
#include "feature/feature.h"
#include "modulated/modulated.h"
#include "beamformer/beamformer.h"

namespace {

struct Graph {
  std::vector<SampleFeaturePtr> samples;
  std::vector<OverSampledDFTAnalysisBankPtr> banks;
  SubbandGSCPtr bf;
  OverSampledDFTSynthesisBankPtr syn;
};

// int16-scale integer noise + a common component, the same recipe for every run (values do not matter for the timing; they are
// integers because that is what SampleFeature delivers, feature/feature.cc:265-269)
void fill_pcm(std::vector<float>& x, unsigned seed)
{
  unsigned s = seed * 2654435761u + 12345u;
  for (size_t i = 0; i < x.size(); i++) {
    s = s * 1664525u + 1013904223u;
    x[i] = (float)((int)((s >> 16) & 0x7ff) - 1024);
  }
}

double now_s()
{
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace

int main(int argc, char** argv)
{
  if (argc != 10) { fprintf(stderr, "usage: %s coeffs.f64 M m r channels frames graphs block_frames pool\n", argv[0]); return 2; }
  const unsigned M = atoi(argv[2]), m = atoi(argv[3]), r = atoi(argv[4]), N = atoi(argv[5]);
  const long frames = atol(argv[6]), block_frames = atol(argv[8]);
  const int G = atoi(argv[7]), use_pool = atoi(argv[9]);
  const unsigned D = M >> r;
  try {
    gsl_vector* h_fb = gsl_vector_calloc(m * M);
    gsl_vector* g_fb = gsl_vector_calloc(m * M);
    FILE* fc = fopen(argv[1], "rb");
    if (!fc || fread(h_fb->data, sizeof(double), m * M, fc) != m * M || fread(g_fb->data, sizeof(double), m * M, fc) != m * M) {
      fprintf(stderr, "cannot read %s\n", argv[1]); return 2;
    }
    fclose(fc);
    gsl_vector* delays = gsl_vector_calloc(N);
    for (unsigned c = 0; c < N; c++) gsl_vector_set(delays, c, ((double)c - N / 2) * 20.0 * 0.26 / 343740.0);   // 20 mm line array, ~75 deg
    std::vector<Graph> graphs(G);
    std::vector<float> pcm((size_t)frames * D);
    for (int g = 0; g < G; g++) {
      Graph& gr = graphs[g];
      gr.bf = new SubbandGSC(M, false);
      for (unsigned c = 0; c < N; c++) {
        SampleFeaturePtr sf = new SampleFeature("", D, D, true);
        OverSampledDFTAnalysisBankPtr bank = new OverSampledDFTAnalysisBank((VectorFloatFeatureStreamPtr&)sf, h_fb, M, m, r);
        bank->set_block_frames(block_frames);
        gr.bf->setChannel((VectorComplexFeatureStreamPtr&)bank);
        gr.samples.push_back(sf); gr.banks.push_back(bank);
      }
      gr.bf->calcGSCWeights(16000.0, delays);
      gr.syn = new OverSampledDFTSynthesisBank((VectorComplexFeatureStreamPtr&)gr.bf, g_fb, M, m, r);
    }
    SubbandGraphPoolPtr pool;
    if (use_pool) {
      pool = new SubbandGraphPool();
      for (int g = 0; g < G; g++) pool->add((SubbandDSPtr&)graphs[g].bf, graphs[g].syn);
    }
    auto load = [&](int pass) {
      for (int g = 0; g < G; g++)
        for (unsigned c = 0; c < N; c++) {
          fill_pcm(pcm, (unsigned)(pass * 7919 + g * 131 + c));
          graphs[g].samples[c]->set_samples(pcm.data(), pcm.size());
        }
    };
    double wall = 0.0, pull = 0.0, upload = 0.0, device = 0.0, checksum = 0.0;
    long blocks = 0, dev_allocs = 0, pin_allocs = 0;
    // pass 0 warms up (stream creation, kernel loading, every buffer at its final size); pass 1 is timed
    for (int pass = 0; pass < 2; pass++) {
      load(pass);
      if (use_pool) pool->reset(); else for (int g = 0; g < G; g++) graphs[g].syn->reset();
      // reset() rewinds the SampleFeatures (samples stay); timing starts with the first pull
      long d0, p0;
      btk_node_alloc_counts(&d0, &p0);
      btk_node_timers_reset();
      blocks = 0; checksum = 0.0;
      const double t0 = now_s();
      if (use_pool) {
        while (pool->next())
          for (int g = 0; g < G; g++) {
            const gsl_vector_float* blk = pool->output(g);
            if (blk) { checksum += gsl_vector_float_get(blk, D / 2); blocks++; }
          }
      } else {
        for (int g = 0; g < G; g++)
          for (;;) {
            const gsl_vector_float* blk;
            try { blk = graphs[g].syn->next(); } catch (jiterator_error&) { break; }
            checksum += gsl_vector_float_get(blk, D / 2); blocks++;
          }
      }
      wall = now_s() - t0;
      btk_node_timers(&pull, &upload, &device);
      long d1, p1;
      btk_node_alloc_counts(&d1, &p1);
      dev_allocs = d1 - d0; pin_allocs = p1 - p0;
    }
    // (a pool stages a round -- pulling included -- inside its upload timer: the pull time is then counted twice)
    double serve = wall - pull - upload - device;
    if (serve < 0) serve = wall - upload - device;
    if (serve < 0) serve = 0;
    printf("{\"graphs\": %d, \"pool\": %d, \"channels\": %u, \"M\": %u, \"frames_per_graph\": %ld, \"block_frames\": %ld, "
           "\"output_blocks\": %ld, \"wall_s\": %.6f, \"frames_per_s\": %.1f, \"pull_sources_s\": %.6f, \"upload_s\": %.6f, "
           "\"device_and_wait_s\": %.6f, \"serve_next_s\": %.6f, \"serve_us_per_frame\": %.3f, \"hipMalloc_in_timed_pass\": %ld, "
           "\"hipHostMalloc_in_timed_pass\": %ld, \"rounds\": %ld, \"checksum\": %.3f}\n",
           G, use_pool, N, M, frames, block_frames, blocks, wall, blocks / wall, pull, upload, device, serve,
           1e6 * serve / (blocks ? blocks : 1), dev_allocs, pin_allocs, use_pool ? pool->rounds() : -1L, checksum);
    gsl_vector_free(h_fb); gsl_vector_free(g_fb); gsl_vector_free(delays);
  } catch (j_error& e) {
    fprintf(stderr, "j_error: %s\n", e.what());
    return 1;
  }
  return 0;
}
