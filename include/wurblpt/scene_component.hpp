/* scene_component.hpp -- base class of everything a Scene owns (reference scene_component.hpp:39-101) */
#pragma once

namespace WurblPT {

class SceneComponent
{
public:
    virtual ~SceneComponent() {}
};

}
