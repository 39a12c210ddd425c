// OptimizerLocal (better_flow_amd/host/better_flow/optimizer_sampler.h) driven the way a caller of the
// reference's class would: both constructors, run(), one explicit score evaluation.  Prints full-precision
// numbers; tests/test_host_cli.py runs it against the oracle shim (CPU) and against libbf_accel.so (GPU)
// and expects the same text.
#include <better_flow/common.h>
#include <better_flow/event_file.h>
#include <better_flow/optimizer_sampler.h>
#include <cstdio>

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    LinearEventCloud cloud;
    EventFile::from_file(&cloud, argv[1]);
    {
        OptimizerLocal ol(&cloud, 3);
        int rc = ol.run();
        std::printf("cloud rc=%d nx=%.17g ny=%.17g score=%.17g evals=%lld\n", rc, ol.get_nx(), ol.get_ny(),
                    ol.get_last_score(), ol.get_evaluations());
        double s0 = ol.iteration_step(0.0, 0.0, true);
        unsigned long long sum = 0;
        const bf::Image2D<uint8_t> &img = ol.get_project_img();
        for (size_t i = 0; i < img.data.size(); ++i) sum += img.data[i] * (unsigned long long)(i % 251 + 1);
        std::printf("cloud score0=%.17g img=%dx%d checksum=%llu\n", s0, img.rows, img.cols, sum);
    }
    {
        Event ec(90, 120, 20000000);
        OptimizerLocal ol(&cloud, ec, 3, 40);
        double sc = ol.iteration_step(0.1, -0.2);
        int rc = ol.run();
        std::printf("window score=%.17g rc=%d nx=%.17g ny=%.17g evals=%lld\n", sc, rc, ol.get_nx(), ol.get_ny(),
                    ol.get_evaluations());
    }
    return 0;
}
