// volly_headless.cpp -- the reference's main loop (main.cpp:293-362) without SDL:
// load one mesh into a Vol_LOD_Set, walk the mouse across the window, render every
// frame on the GPU through volly::State::raycastOntoScreen and print the frame time
// and "raycasts per second" line the reference prints (main.cpp:355-358).  The last
// frame is written as a binary PPM (r', g', b' = bytes 1..3 of each pixel).
//
//   volly_headless <mesh.ply|.off> <curLoD 0..3> <frames> <out.ppm> [mouseX mouseY camX camY camZ]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../volume-renderer_amd/host/volly_state.h"

int main(int argc, char **argv)
{
    if (argc < 5) {
        std::fprintf(stderr, "usage: %s mesh curLoD frames out.ppm [mouseX mouseY camX camY camZ]\n", argv[0]);
        return 2;
    }
    try {
        volly::State state;
        // VOLLY_DEVICE_VOXELIZE=1: build the four grids of the Vol_LOD_Set on the GPU as well
        const char *dv = std::getenv("VOLLY_DEVICE_VOXELIZE");
        if (dv && std::atoi(dv)) state.initAllVolumesOnDevice({argv[1]});
        else state.initAllVolumes({argv[1]});
        state.curLoD = std::atoi(argv[2]);
        const int frames = std::atoi(argv[3]);
        state.mouseX = argc > 5 ? std::atoi(argv[5]) : 0;
        state.mouseY = argc > 6 ? std::atoi(argv[6]) : 0;
        if (argc > 9) state.camPos = volly::vec3((float)std::atof(argv[7]), (float)std::atof(argv[8]), (float)std::atof(argv[9]));
        const int W = state.renderTargetSizeX, H = state.renderTargetSizeY;
        volly::gpu::Texture screenTex(W, H);               // the reference's screenTex (main.h:113), minus SDL
        uint32_t *pixels = screenTex.data_stream;
        for (int f = 0; f < frames; ++f) {
            auto t0 = std::chrono::steady_clock::now();
            state.resolveUserInput();
            state.raycastOntoScreen(pixels);
            double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            std::printf("Frame : %d\n\tCur Frame Time: %.3f\n\tCasting %d*%d rays per frame, this is %.0f raycasts per second\n"
                        "\ton a %d*%d*%d dataset.\n", state.frame, ms, W, H, W * H * (1000.0 / ms),
                        state.curVolSize().x, state.curVolSize().y, state.curVolSize().z);
            if (f + 1 < frames) state.mouseX += 7;
        }
        FILE *fp = std::fopen(argv[4], "wb");
        if (!fp) { std::perror(argv[4]); return 1; }
        std::fprintf(fp, "P6\n%d %d\n255\n", W, H);
        for (size_t i = 0; i < (size_t)W * H; ++i) {
            const uint32_t p = pixels[i];
            unsigned char rgb[3] = {(unsigned char)(p >> 8), (unsigned char)(p >> 16), (unsigned char)(p >> 24)};
            std::fwrite(rgb, 1, 3, fp);
        }
        std::fclose(fp);
    } catch (const std::exception &e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
