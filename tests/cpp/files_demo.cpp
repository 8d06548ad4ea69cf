// Host-only check of include/mtr_files.hpp: parse the resource files named on the command line and print what the
// reference's readers would log.  usage: files_demo model.mod shader.mfx material.mrl texture.tex schedule.sdl
#include "mtr_files.hpp"
#include <cstdio>

int main(int argc, char** argv) {
    if (argc != 6) return 64;
    try {
        mtr::ModelFile mf(mtr::read_file(argv[1]));
        mtr::Shader2File sh(mtr::read_file(argv[2]));
        mtr::MaterialFile mat(mtr::read_file(argv[3]), sh);
        mtr::TextureFile tex(mtr::read_file(argv[4]));
        mtr::SchedulerFile sdl(mtr::read_file(argv[5]));
        std::printf("model prims=%u materials=%u boundaries=%u\n", mf.view().primitive_num, mf.view().material_num, mf.view().boundary_num);
        for (uint32_t p = 0; p < mf.view().primitive_num; p++) {
            const uint32_t handle = mf.primitive_field(p, MTR_PRIM_INPUTLAYOUT);
            const int32_t obj = sh.get_object_by_handle(handle);
            uint32_t stride = 0;
            const mtr_layout l = sh.input_layout((uint32_t)obj, &stride);
            const std::string mname = mf.material_name(mf.primitive_field(p, MTR_PRIM_MATERIAL_NO));
            const int32_t mi = mat.material_by_name(mname);
            std::printf("prim %u layout=%s stride=%u bound=%u material=%s albedo=%d joint=%u\n", p, sh.object_name((uint32_t)obj).c_str(), stride,
                        l.num_elements, mname.c_str(), mi < 0 ? -1 : mat.material((uint32_t)mi).albedo_texture,
                        mf.boundary_joint(mf.primitive_field(p, MTR_PRIM_BOUNDARY_NUM)));
        }
        std::printf("texture %ux%u format=%u\n", tex.width(), tex.height(), tex.format());
        for (uint32_t t = 0; t < sdl.num_tracks(); t++) std::printf("track %u type=%u keys=%u name=%s\n", t, sdl.track(t).track_type, sdl.track(t).key_num, sdl.track(t).name);
        std::printf("eval bool@31=%llu\n", (unsigned long long)sdl.eval(2, 31));
        return 0;
    } catch (const mtr::Error& e) {
        std::fprintf(stderr, "%s\n", e.what());
        return 2;
    }
}
