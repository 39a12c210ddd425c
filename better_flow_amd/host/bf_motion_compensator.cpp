// bf_motion_compensator -- command-line front end (mirror of the reference's
// better_flow_core/src/bf_motion_compensator.cpp:61-216): same flags, same text input
// ("t x y p") and the same "-o" output ("t x y 1 v u"), with the optimizer on the MI355X.
//
// Additional flags (the reference fixes these at compile time, common.h:39-40 /
// bf_motion_compensator.cpp:6-7): --res-x= --res-y= (sensor rows / columns), --scale=,
// --max-iter=, --device=.
#include <better_flow/common.h>
#include <better_flow/dvs_flow.h>

#define EVENT_WIDTH 50000
#define TIME_WIDTH 0.2

float time_refresh = 0.033;
unsigned long long int event_refresh = 20000;

bool manual = false;
bool quiet = false;
char *file = NULL;
char *outFileName = NULL;
const char *to_bin = NULL;
bool gpu = false;
bool img = false;
bool video = false;
bool stm_disable = false;
bool bufferize_file = false;
std::string img_prefix = "./";
std::string video_name = "./out.avi";
int video_fps = 60;
int opt_scale = 3;
int opt_max_iter = -1;

static void lPrintVersion() {
    printf("DVS flow estimator (better flow), %s (build %s @ %s)\n", BF_VERSION, __DATE__, __TIME__);
    printf("\tCompiled with maximum event memory of %i events\n\tand slice size of %f seconds.\n", EVENT_WIDTH,
           TIME_WIDTH);
    printf("\tMotion compensation runs on the GPU: %s\n", bf_version());
}

static void usage(int ret) {
    lPrintVersion();
    printf("\nusage: bf_motion_compensator\n");
    printf("    [--refresh-time={0.0 - inf}]\t\tRun processing when at least this amount of time (floatimg point,\n");
    printf("                                \t\tseconds) has passed since the last processing, (default = %f)\n", time_refresh);
    printf("    [--refresh-event-count={0 - inf}]\t\tRun processing when at least this number of new events has\n");
    printf("                                     \t\tarrived since the last processing (default = %llu)\n", event_refresh);
    printf("    [-i/--interactive]\tEnable interactive mode\n");
    printf("    [-G]\t\t\t\tUse GPU support (always on in this build)\n");
    printf("    [--stm-disable]\t\t\t\tDo not use previous estimate as a starting point for a new estimate\n");
    printf("    [--img]\t\t\t\tOutput flow images after every iteration\n");
    printf("    [--img-prefix <name>]\t\t\t\tSpecify prefix for the generated image files (default = %s)\n", img_prefix.c_str());
    printf("    [--video]\t\t\t\tOutput a video with flow frames\n");
    printf("    [--video-name <name>]\t\t\t\tSpecify the name of the video file (default = %s)\n", video_name.c_str());
    printf("    [--video-fps=<value>]\t\t\t\tSpecify video framerate (default = %i)\n", video_fps);
    printf("    [--bufferize-file]\t\t\t\tRead input file to the buffer first (useful for performance testing)\n");
    printf("    [--quiet]\t\t\t\tSuppress all output\n");
    printf("    [-o <name>/--outfile=<name>]\tOutput filename (may be \"-\" for standard output)\n");
    printf("    [--version]\t\t\t\tPrint better flow version\n");
    printf("    [--res-x=<rows>] [--res-y=<columns>]\tSensor size (default = %d x %d)\n", RES_X, RES_Y);
    printf("    [--scale=<odd>]\t\t\t\tImage scale used by the minimizer (default = %d)\n", opt_scale);
    printf("    [--max-iter=<n>]\t\t\t\tCap on minimizer iterations per slice (default = unlimited)\n");
    printf("    [--device=<n>]\t\t\t\tHIP device index (default = 0)\n");
    printf("    [--to-bin=<name>]\t\t\t\tOnly convert the (text) input to the binary event format and exit\n");
    printf("    <file to process or \"-\" for stdin>\n");
    exit(ret);
}

int main(int argc, char *argv[]) {
    if (argc == 1) usage(1);
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--help"))
            usage(0);
        else if (!strcmp(argv[i], "-v") || !strcmp(argv[i], "--version")) {
            lPrintVersion();
            return 0;
        } else if (!strcmp(argv[i], "--quiet"))
            quiet = true;
        else if (!strncmp(argv[i], "--refresh-time=", 15))
            time_refresh = atof(argv[i] + 15);
        else if (!strncmp(argv[i], "--refresh-event-count=", 22))
            event_refresh = atoi(argv[i] + 22);
        else if (!strcmp(argv[i], "-G"))
            gpu = true;
        else if (!strcmp(argv[i], "-i"))
            manual = true;
        else if (!strcmp(argv[i], "--interactive"))
            manual = true;
        else if (!strcmp(argv[i], "--bufferize-file"))
            bufferize_file = true;
        else if (!strcmp(argv[i], "--stm-disable"))
            stm_disable = true;
        else if (!strcmp(argv[i], "--img"))
            img = true;
        else if (!strcmp(argv[i], "--img-prefix")) {
            if (++i == argc) {
                fprintf(stderr, "No output file specified after --img-prefix option.\n");
                usage(1);
            }
            img_prefix = argv[i];
        } else if (!strcmp(argv[i], "--video"))
            video = true;
        else if (!strcmp(argv[i], "--video-name")) {
            if (++i == argc) {
                fprintf(stderr, "No output file specified after --video-name option.\n");
                usage(1);
            }
            video_name = argv[i];
        } else if (!strncmp(argv[i], "--video-fps=", 12))
            video_fps = atoi(argv[i] + 12);
        else if (!strncmp(argv[i], "--res-x=", 8))
            bf::sensor().res_x = atoi(argv[i] + 8);
        else if (!strncmp(argv[i], "--res-y=", 8))
            bf::sensor().res_y = atoi(argv[i] + 8);
        else if (!strncmp(argv[i], "--scale=", 8))
            opt_scale = atoi(argv[i] + 8);
        else if (!strncmp(argv[i], "--max-iter=", 11))
            opt_max_iter = atoi(argv[i] + 11);
        else if (!strncmp(argv[i], "--device=", 9))
            bf::DeviceContext::device() = atoi(argv[i] + 9);
        else if (!strncmp(argv[i], "--to-bin=", 9))
            to_bin = argv[i] + 9;
        else if (!strcmp(argv[i], "-o")) {
            if (++i == argc) {
                fprintf(stderr, "No output file specified after -o option.\n");
                usage(1);
            }
            outFileName = argv[i];
        } else if (!strncmp(argv[i], "--outfile=", 10))
            outFileName = argv[i] + strlen("--outfile=");
        else if (!strcmp(argv[i], "-")) {
        } else if (argv[i][0] == '-') {
            fprintf(stderr, "Unknown option \"%s\".\n", argv[i]);
            usage(1);
        } else {
            if (file != NULL) {
                fprintf(stderr, "Multiple input files specified on command line: \"%s\" and \"%s\".\n", file, argv[i]);
                usage(1);
            } else
                file = argv[i];
        }
    }
    if (file == NULL) {
        fprintf(stderr, "No input file.\n");
        usage(1);
    }
    if (to_bin != NULL) {   // text -> binary structure-of-arrays (better_flow/event_reader.h); no GPU involved
        bf::EventReader reader(file);
        if (!reader.good()) {
            fprintf(stderr, "cannot read '%s'\n", file);
            return 1;
        }
        std::vector<uint64_t> t;
        std::vector<uint16_t> x, y;
        std::vector<uint8_t> p;
        reader.for_each_event([&](unsigned row, unsigned col, unsigned long long t_ns) {
            t.push_back(t_ns); x.push_back((uint16_t)col); y.push_back((uint16_t)row); p.push_back(1);
        });
        if (!bf::EventReader::write_binary(to_bin, t, x, y, p)) {
            fprintf(stderr, "cannot write '%s'\n", to_bin);
            return 1;
        }
        if (!quiet) std::cout << "Converted " << t.size() << " events to " << to_bin << std::endl;
        return 0;
    }
    if (opt_scale < 1 || opt_scale % 2 == 0) {
        fprintf(stderr, "--scale must be odd.\n");
        return 1;
    }

    DVS_flow<EVENT_WIDTH, (sll)FROM_SEC(TIME_WIDTH)> estimator(event_refresh, FROM_SEC(time_refresh));
    estimator.set_quiet(quiet);
    estimator.set_scale(opt_scale);
    estimator.set_max_iter(opt_max_iter);
    if (outFileName != NULL) estimator.set_accumulate();   // This will enable event bufferization
    if (manual) estimator.set_manual_mode(true);
    if (img) estimator.set_generate_pictures(true, img_prefix);
    if (video) estimator.set_generate_video(true, video_name, video_fps);
    if (stm_disable) estimator.set_stm_disable(true);

    if (bufferize_file) {   // read the input file to the buffer first
        LinearEventCloud ec;
        EventFile::from_file(&ec, file);
        clock_t begin = std::clock();
        clock_t begin_slice = std::clock();
        ull i = 0;
        for (auto &e : ec) {
            ++i;
            bool processed = estimator.add_event(e);
            if (processed) {
                clock_t end_slice = std::clock();
                if (!quiet)
                    std::cout << float(i * 100) / float(ec.size()) << " %\t" << i << "\t"
                              << (double(end_slice - begin_slice) / CLOCKS_PER_SEC) << " sec\t"
                              << estimator.get_buf_size() << " events\t"
                              << double(estimator.get_time_diff()) / 1000000000.0 << " slice_td\t"
                              << double(estimator.get_buf_time_diff()) / 1000000000.0 << " buffer_td\n";
                begin_slice = std::clock();
            }
        }
        clock_t end = std::clock();
        std::cout << "Toatal flow elapsed: " << double(end - begin) / CLOCKS_PER_SEC << " sec." << std::endl << std::flush;
    } else {
        if (!quiet) std::cout << "Reading from file... (" << file << ")" << std::endl << std::flush;
        bf::EventReader reader(file);   // text "t x y p" or binary SoA (better_flow/event_reader.h)
        if (!reader.good()) {
            fprintf(stderr, "cannot read '%s'\n", file);
            return 1;
        }
        ull i = reader.for_each_event([&](unsigned row, unsigned col, unsigned long long t_ns) {
            Event e(row, col, (ull)t_ns);
            estimator.add_event(e);
        });
        if (!quiet) std::cout << "Read and processed " << i << " events" << std::endl << std::flush;
    }

    estimator.recompute();   // Ensure that *every* event has been processed

    if (outFileName != NULL) {
        LinearEventCloudTemplate<Event> accumulated = estimator.get_accumulated();
        EventFile::to_file_uv(&accumulated, outFileName);
    }
    if (!quiet)
        std::cout << "slices: " << estimator.get_slices_done() << " (skipped " << estimator.get_slices_skipped()
                  << "), minimizer iterations: " << estimator.get_iterations_total() << std::endl;
    bf::DeviceContext::release();
    return 0;
}
